cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06s; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())" > /dev/null 2>&1
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), "lat", round(j["latency_ms_single_proof"], 2), j["config"]["results_sha256"][:12])'
for rep in 1 2; do
  for n in 1 2 4; do
    python bench.py $C --ctx-option msm_sort_streams=$n 2>/dev/null | python -c "$P" sort_streams=$n | tee -a $O/bench.txt
  done
done
for w in plonk marlin; do
  for n in 1 4; do
    tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 --ctx-option msm_sort_streams=$n 2>&1 | grep -o '"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"output_sha256": "[0-9a-f]\{12\}' | tr '\n' ' ' | sed "s/^/$w alone streams=$n /" | tee -a $O/poly.txt; echo | tee -a $O/poly.txt
    tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 --ctx-option msm_sort_streams=$n 2>&1 | grep -o '"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"output_sha256": "[0-9a-f]\{12\}' | tr '\n' ' ' | sed "s/^/$w 4inflight streams=$n /" | tee -a $O/poly.txt; echo | tee -a $O/poly.txt
  done
done
python bench.py --log-n 22 --steps 4 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report --ctx-option msm_sort_streams=1 2>/dev/null | python -c "$P" 2e22_streams=1 | tee -a $O/bench.txt
python bench.py --log-n 22 --steps 4 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report --ctx-option msm_sort_streams=4 2>/dev/null | python -c "$P" 2e22_streams=4 | tee -a $O/bench.txt

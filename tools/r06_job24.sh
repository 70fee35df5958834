cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06x; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "same_scalars or msm" > $O/test_msm.txt 2>&1; tail -5 $O/test_msm.txt
timeout 900 python -m pytest tests/test_chaos.py tests/test_marks.py tests/test_abi.py -m gpu -x -q > $O/test_chaos.txt 2>&1; tail -5 $O/test_chaos.txt
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), "lat", round(j["latency_ms_single_proof"], 2), {k: round(v, 1) for k, v in j["stream_elapsed_ms_per_step"].items() if k != "note"}, j["config"]["results_sha256"][:12])'
LAB=$PWD/collaborative-zksnark_amd/libczk_hip_lab.so
for rep in 1 2 3; do
  python bench.py $C --ctx-option msm_sort_reuse=0 2>/dev/null | python -c "$P" product_reuse=0 | tee -a $O/bench.txt
  python bench.py $C --ctx-option msm_sort_reuse=1 2>/dev/null | python -c "$P" product_reuse=1 | tee -a $O/bench.txt
  CZK_LIB_PATH=$LAB python bench.py $C --ctx-option msm_sort_reuse=0 2>/dev/null | python -c "$P" lab_reuse=0 | tee -a $O/bench.txt
  CZK_LIB_PATH=$LAB python bench.py $C --ctx-option msm_sort_reuse=1 --ctx-option msm_sort_reuse_any_inf=1 2>/dev/null | python -c "$P" lab_bound_any_inf | tee -a $O/bench.txt
done

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "witness or fused" > $O/test_fused.txt 2>&1; tail -8 $O/test_fused.txt
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), "lat", round(j["latency_ms_single_proof"], 2), {k: round(v, 1) for k, v in j["stream_elapsed_ms_per_step"].items() if k != "note"}, j["config"]["results_sha256"][:12])'
for rep in 1 2 3; do
  for f in 0 1; do
    python bench.py $C --ctx-option ntt_fuse_pairs=$f 2>/dev/null | python -c "$P" fuse=$f | tee -a $O/bench.txt
  done
done
python tools/ntt_bench.py 21 2>/dev/null | tail -4

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06o; mkdir -p $O
AB=$PWD/collaborative-zksnark_amd/libczk_hip_base.so
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), {k: round(v, 1) for k, v in j["stream_elapsed_ms_per_step"].items() if k != "note"}, j["config"]["results_sha256"][:12])'
for rep in 1 2; do
  CZK_LIB_PATH=$AB python bench.py $C 2>/dev/null | python -c "$P" base | tee -a $O/bench.txt
  python bench.py $C 2>/dev/null | python -c "$P" small64 | tee -a $O/bench.txt
done

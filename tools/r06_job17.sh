cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06q; mkdir -p $O
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), "lat", round(j["latency_ms_single_proof"], 2), j["config"]["results_sha256"][:12])'
for rep in 1 2; do
  for o in b_g2,l,a,b_g1 l,a,b_g1,b_g2 l,a,b_g2,b_g1 l,b_g2,a,b_g1; do
    python bench.py $C --msm-order $o 2>/dev/null | python -c "$P" $o | tee -a $O/bench.txt
  done
done

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06k; mkdir -p $O; rm -f $O/*
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = j.get("contexts_in_flight") or {}; print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "contexts", c.get("contexts"), round(c.get("proofs_per_s", 0), 3), round(c.get("vs_value", 0), 4), c.get("error"))'
for q in default 8 16; do
  for k in 3; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-seam-report --no-verify-report --no-other-workloads --contexts-report $k 2>$O/err_$q.txt | python -c "$P" "hw_queues=$q" | tee -a $O/contexts.txt
  done
done

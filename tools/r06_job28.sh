cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06c; mkdir -p $O; rm -f $O/*
C="--steps 80 --warmup 5 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3))'
python bench.py $C 2>/dev/null | python -c "$P" solo | tee -a $O/corun.txt
for n in 2 3; do
  for k in $(seq 1 $n); do
    ( python bench.py $C 2>/dev/null | python -c "$P" "concurrent_${n}_proc_$k" >> $O/corun.txt ) &
  done
  wait
done
python bench.py $C 2>/dev/null | python -c "$P" solo | tee -a $O/corun.txt
cat $O/corun.txt

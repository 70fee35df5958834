#!/bin/bash
# kernel trace of the Plonk / Marlin workloads + what runs while no accumulate kernel does (tools/timeline_gaps.py)
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/${TAG}_polyiop
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in "plonk --parties 3 --log-n 18" "marlin --parties 2 --log-n 20"; do
  name=$(echo $w | cut -d' ' -f1)
  rocprofv3 --kernel-trace -d $OUT/trace_$name -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/trace_$name.log
  DB=$(find $OUT/trace_$name -name '*.db' | head -1)
  python $GRAFT_REPO_ROOT/tools/timeline_gaps.py $DB > $OUT/gaps_$name.txt 2>&1
  cat $OUT/gaps_$name.txt
  python -c "import json,sys; j=json.load(open('$OUT/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['accumulate_busy_frac'])"
  find $OUT/trace_$name -name '*.db' -size +30M -delete
done

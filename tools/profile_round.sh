#!/bin/bash
# One GPU-box pass that regenerates the round's judged profiles (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command, three separate PMC passes (HBM traffic, instruction counts / clock; per-kernel share of a proof's VALU instructions), the stage
#   and width micro-benchmarks.  Raw databases stay under gpurun_out/; the summaries are copied to profiles/ by hand afterwards.
#   usage: bash tools/profile_round.sh r03
set -u
R=${1:-r03}
OUT=$PWD/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
PMC="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seam-report --no-other-workloads --no-result-check --no-verify-report"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o pmc -- $PMC > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.log
done
cd - > /dev/null
DB=$(find $OUT/trace -name '*.db' | head -1)
python tools/prof_summary.py $DB 20 > $OUT/kernel_trace_stats.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_traffic.json $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU -name '*.db') > $OUT/pmc_summary.log 2>&1
python tools/valu_share.py $(find $OUT/pmc_SQ_INSTS_VALU -name '*.db' | head -1) > $OUT/valu_share.txt 2>&1
python tools/stage_bench.py 20 4 > $OUT/stage_bench.txt 2>&1
python tools/width_bench.py 3 > $OUT/width_bench.txt 2>&1
ls -la $OUT | head -30
tail -5 $OUT/kernel_trace_stats.txt

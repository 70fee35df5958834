#!/bin/bash
# Round-5 GPU-box pass (run through gpurun from the repo root): FETCH_SIZE calibration, the default bench line, Plonk / Marlin alone and pipelined with
# their timeline gaps, the single-thread CPU baseline at full size, then tools/profile_round.sh.  Summaries are copied to profiles/ by hand.
set -u
R=$PWD
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib -o calib -- $R/tools/fetch_calib.bin 4 64 > $OUT/fetch_calib.txt 2> $OUT/fetch_calib.log
python $R/tools/fetch_calib_summary.py $(find $OUT/calib -name '*.db' | head -1) $OUT/fetch_calib.txt > $OUT/fetch_calib.json 2>> $OUT/fetch_calib.log
cat $OUT/fetch_calib.json
cd $R
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 1500 $OUT/bench_default.json
for w in "plonk --parties 3 --log-n 18" "marlin --parties 2 --log-n 20"; do
  name=$(echo $w | cut -d' ' -f1)
  python bench.py --workload $w --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/bench_${name}_alone.json 2> $OUT/bench_${name}_alone.err
  python bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  python -c "import json; a=json.load(open('$OUT/bench_${name}_alone.json')); b=json.load(open('$OUT/bench_${name}.json')); print('$name alone', a['ms_per_step'], 'pipelined', b['value'], b['ms_per_step'], b['accumulate_busy_frac'])"
done
bash tools/run_polyiop_trace.sh r05 > $OUT/polyiop_trace.txt 2>&1
tail -45 $OUT/polyiop_trace.txt
python tools/cpu_baseline_validate.py $OUT/cpu_baseline_validation.json 14 20 > $OUT/cpu_baseline_validation.log 2>&1
cat $OUT/cpu_baseline_validation.log
bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1
tail -12 $OUT/profile_round.log

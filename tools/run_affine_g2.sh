#!/bin/bash
# gpurun driver for tools/affine_bench_g2.bin: timing, big-integer check, instruction counts (separate PMC pass)
set -u
OUT=$PWD/gpurun_out/r04_affine_g2
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/tools/affine_bench_g2.bin
timeout 300 $BIN /tmp/affine_g2_dump.bin > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
timeout 600 python $PWD/tools/affine_check_g2.py /tmp/affine_g2_dump.bin > $OUT/check.txt 2>&1
cat $OUT/check.txt
rm -f /tmp/affine_g2_dump.bin
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc -o pmc -- $BIN > $OUT/pmc_run.txt 2> $OUT/pmc.log
cd - > /dev/null
DB=$(find $OUT/pmc -name '*.db' | head -1)
python - "$DB" > $OUT/pmc_summary.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
q = ("select counter_name, kernel_name, count(*), sum(value), avg(duration) from counters_collection "
     "group by counter_name, kernel_name order by counter_name, kernel_name")
for r in db.execute(q):
    print(r)
PY
cat $OUT/pmc_summary.txt
find $OUT/pmc -name '*.db' -size +20M -delete

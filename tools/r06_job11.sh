# why does lane interleaving help?  PMC passes of the isolated MSM stages, interleave 1 vs 4 (tools/stage_bench.py 20 4)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06k; mkdir -p $O
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
python tools/stage_bench.py 20 4 msm_lane_interleave=1 > $O/stage_il1.txt 2>&1
python tools/stage_bench.py 20 4 msm_lane_interleave=4 > $O/stage_il4.txt 2>&1
python tools/stage_bench.py 20 4 msm_lane_interleave=1 >> $O/stage_il1.txt 2>&1
python tools/stage_bench.py 20 4 msm_lane_interleave=4 >> $O/stage_il4.txt 2>&1
cat $O/stage_il1.txt $O/stage_il4.txt
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_sum TCC_EA0_RDREQ_IO_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  for g in 1 4; do
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$g -o pmc -- python $GRAFT_REPO_ROOT/tools/stage_bench.py 20 4 msm_lane_interleave=$g > /dev/null 2> /tmp/pmc_${tag}_$g.log
    DB=$(find /tmp/pmc_${tag}_$g -name '*.db' | head -1)
    if [ -n "$DB" ]; then python - "$DB" "$g" <<'PY' >> $O/pmc_compare.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
try:
    for name, kernel, n, avg, dur in db.execute("select counter_name, kernel_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%k_accumulate_te%' or kernel_name like '%k_accumulate_u2%' group by counter_name, kernel_name order by kernel_name, counter_name"):
        print(f"il={sys.argv[2]} {kernel[:28]:28s} {name:34s} n={n} avg={avg:.4g} dur_ms={dur/1e6:.3f}")
except Exception as e:
    print("il", sys.argv[2], "ERR", e)
PY
    else tail -3 /tmp/pmc_${tag}_$g.log >> $O/pmc_compare.txt; fi
  done
done
sort -k2,3 -s $O/pmc_compare.txt | head -120

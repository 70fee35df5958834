# round 6: Plonk shortcuts in the compiled host; the two bounded kernel experiments of VERDICT r05 item 4 (lab library, A/B/A/B on one box)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
for w in plonk marlin; do
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone.json 2>&1
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone_b.json 2>&1
  tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 > $O/${w}_4inflight.json 2>&1
  tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 2 --breakdown > $O/${w}_breakdown.json 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  TIMELINE_DUMP=$O/${w}_kernels.txt python tools/proof_timeline.py $DB 5 > $O/${w}_timeline.txt 2>&1
done
grep -h -o '"workload": "[a-z]*".*"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"latency_ms_single_proof": [0-9.]*\|"ntt_lanes_per_proof": [0-9.]*\|"output_sha256": "[0-9a-f]*"' $O/*.json
timeout 900 python -m pytest tests/test_pipelines.py tests/test_gpu_parity.py -m gpu -x -q -k 'pipelines or cpp or lincomb or evaluate_many or div_vanishing' > $O/test_pipes.txt 2>&1; tail -3 $O/test_pipes.txt
# --- experiments: lab library, default bench (Groth16 configs[1]), alternating
LAB=$PWD/collaborative-zksnark_amd/libczk_hip_lab.so
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
for rep in 1 2; do
  CZK_LIB_PATH=$LAB $B > $O/g16_lab_base_$rep.json 2>/dev/null
  CZK_LIB_PATH=$LAB CZK_NTT_SKIP_COSET_FIRST=1 $B --no-result-check > $O/g16_lab_nttskip_$rep.json 2>/dev/null
  CZK_LIB_PATH=$LAB CZK_G1_LANE_PAIRS=1 $B > $O/g16_lab_pairs_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06f/g16_lab_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 3), "proofs/s", round(j["ms_per_step"], 3), "ms", "checked", j.get("results_checked"), "acc_g1_ms", round(j["roofline"]["avg_launch_ms"], 3),
              "ntt_pass", round(j["stream_elapsed_ms_per_step"]["ntt_pass"], 2), "sha", j["config"]["results_sha256"][:12])
    except Exception as e:
        print(f, "ERR", e)
PY
# lane pairs: FETCH_SIZE and the clock of the G1 accumulate kernel, base vs pairs (one proof under --pmc, separate passes)
PMC="python $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-seam-report --no-other-workloads --no-result-check --no-verify-report"
cd /tmp
for v in base pairs; do
  for c in FETCH_SIZE "GRBM_GUI_ACTIVE SQ_WAVES"; do
    tag=$(echo $c | cut -d' ' -f1)
    if [ $v = pairs ]; then export CZK_G1_LANE_PAIRS=1; else unset CZK_G1_LANE_PAIRS; fi
    CZK_LIB_PATH=$LAB rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${v}_$tag -o pmc -- $PMC > /dev/null 2> /tmp/pmc_${v}_$tag.log
    DB=$(find /tmp/pmc_${v}_$tag -name '*.db' | head -1)
    python - "$DB" "$v" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/r06f/pairs_pmc.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, kernel, n, avg, dur in db.execute("select counter_name, kernel_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%k_accumulate_te%' group by counter_name, kernel_name"):
    print(sys.argv[2], name, kernel[:40], "dispatches", n, "avg", avg, "avg_duration_ns", dur)
PY
  done
done
unset CZK_G1_LANE_PAIRS
cat $GRAFT_REPO_ROOT/gpurun_out/r06f/pairs_pmc.txt

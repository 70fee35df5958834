# round 6: msm_lane_interleave beyond 4 lanes and on the table-free path
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O
C="--no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
for g in 1 4 6 8; do
  python bench.py --parties 3 --steps 8 --warmup 2 $C --ctx-option msm_lane_interleave=$g > $O/g16_spdz3_il$g.json 2>$O/err.txt
  python bench.py --parties 4 --log-n 19 --steps 8 --warmup 2 $C --ctx-option msm_lane_interleave=$g > $O/g16_spdz4_2e19_il$g.json 2>$O/err.txt
done
for g in 1 4 13 16 64; do
  python bench.py --no-tables --steps 6 --warmup 2 $C --ctx-option msm_lane_interleave=$g > $O/g16_notables_il$g.json 2>$O/err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06h/g16_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 3), "proofs/s", round(j["ms_per_step"], 3), "ms", "checked", j.get("results_checked"), "lat", round(j["latency_ms_single_proof"], 2))
    except Exception as e:
        print(f, "ERR", e)
PY

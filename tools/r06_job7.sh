# round 6: lane interleaving of the accumulate kernels (msm_lane_interleave), same box, alternating
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm" > $O/test_msm.txt 2>&1; tail -3 $O/test_msm.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
for rep in 1 2; do
  for g in 1 2 4; do
    $B --ctx-option msm_lane_interleave=$g > $O/g16_il${g}_$rep.json 2>$O/err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06g/g16_il*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 3), "proofs/s", round(j["ms_per_step"], 3), "ms", "checked", j.get("results_checked"), "acc_g1_ms", round(j["roofline"]["avg_launch_ms"], 3),
              "acc_g2_ms", round(j["roofline_g2"]["avg_launch_ms"], 3), "power", j.get("power_w_avg") and round(j["power_w_avg"]), "per_kJ", j.get("proofs_per_kJ") and round(j["proofs_per_kJ"], 2), "sha", j["config"]["results_sha256"][:12])
    except Exception as e:
        print(f, "ERR", e)
PY
for w in plonk marlin; do
  for g in 1 2 3 4; do
    tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 --ctx-option msm_lane_interleave=$g > $O/${w}_alone_il$g.json 2>&1
    tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 --ctx-option msm_lane_interleave=$g > $O/${w}_4inflight_il$g.json 2>&1
  done
done
grep -H -o '"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"latency_ms_single_proof": [0-9.]*\|"output_sha256": "[0-9a-f]\{12\}' $O/plonk_*.json $O/marlin_*.json
# other Groth16 shapes: 2^22 (configs[4] size), HBC-2 (2 lanes), SPDZ-3 (6 lanes)
for g in 1 2 4; do
  python bench.py --log-n 22 --steps 4 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report --ctx-option msm_lane_interleave=$g > $O/g16_2e22_il$g.json 2>/dev/null
  python bench.py --scheme hbc --steps 12 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report --ctx-option msm_lane_interleave=$g > $O/g16_hbc2_il$g.json 2>/dev/null
  python bench.py --parties 3 --steps 8 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report --ctx-option msm_lane_interleave=$g > $O/g16_spdz3_il$g.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06g/g16_2e22*.json") + glob.glob("gpurun_out/r06g/g16_hbc2*.json") + glob.glob("gpurun_out/r06g/g16_spdz3*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 3), "proofs/s", round(j["ms_per_step"], 3), "ms", "checked", j.get("results_checked"))
    except Exception as e:
        print(f, "ERR", e)
PY

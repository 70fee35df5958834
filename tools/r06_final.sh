# round 6: what the driver runs at round end, on one box: pytest -m gpu, smoke(), bench.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_tests.txt 2>&1; tail -22 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_final/bench_default.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "latency_ms_single_proof", "results_checked", "power_w_avg", "proofs_per_kJ", "accumulate_busy_frac")})
print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "traffic", "traffic_source")})
print("verify", j.get("proof_verifies", {}).get("proof_verifies"))
for k, v in j.get("other_workloads", {}).items():
    print(k, {a: v.get(a) for a in ("proofs_per_s", "ms_per_proof", "latency_ms_single_proof", "results_checked", "error", "skipped")}, "verifies", (v.get("proof_verifies") or {}).get("proof_verifies") if isinstance(v.get("proof_verifies"), dict) else v.get("proof_verifies"),
          "cpp", {a: (v.get("cpp_host") or {}).get(a) for a in ("proofs_per_s", "latency_ms_single_proof")})
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY

# round 6: stream priorities for one proof alone (Plonk / Marlin from the compiled host), the chaos tests, the power sampler
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
for w in plonk marlin; do
  for prio in 0 1 2; do
    tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 --ctx-option msm_stream_priority=$prio > $O/${w}_alone_prio$prio.json 2>&1
    tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 --ctx-option msm_stream_priority=$prio > $O/${w}_4inflight_prio$prio.json 2>&1
  done
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 --ctx-option msm_stream_priority=1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  python tools/proof_timeline.py $DB 5 > $O/${w}_timeline_prio1.txt 2>&1
done
grep -h -o '"workload": "[a-z]*".*"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"latency_ms_single_proof": [0-9.]*' $O/*.json
timeout 900 python -m pytest tests/test_chaos.py tests/test_marks.py -m gpu -x -q --durations=5 > $O/test_chaos.txt 2>&1; tail -15 $O/test_chaos.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report > $O/bench_power.json 2> $O/bench_power.err; python -c "
import json; j=json.loads(open('$O/bench_power.json').read().strip().splitlines()[-1]); print({k: j.get(k) for k in ('value','ms_per_step','power_w_avg','power_w_max','proofs_per_kJ','power_source','power_samples')})"

# round 6: batched affine conversion, split gates chain, czk_poly_div_vanishing / czk_poly_evaluate_many / czk_fr_lincomb in the compiled host
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "div_vanishing or evaluate_many or lincomb or div_linear" > $O/test_new.txt 2>&1; tail -5 $O/test_new.txt
for w in plonk marlin; do
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone.json 2>&1
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone_b.json 2>&1
  tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 > $O/${w}_4inflight.json 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  TIMELINE_DUMP=$O/${w}_kernels.txt python tools/proof_timeline.py $DB 5 > $O/${w}_timeline.txt 2>&1
done
grep -h -o '"workload": "[a-z]*".*"proofs_in_flight": [0-9]*, "ms_per_proof": [0-9.]*\|"latency_ms_single_proof": [0-9.]*\|"output_sha256": "[0-9a-f]*"' $O/*.json
timeout 1200 python -m pytest tests/test_pipelines.py tests/test_chaos.py tests/test_marks.py tests/test_verify.py -m gpu -x -q > $O/test_pipes.txt 2>&1; tail -5 $O/test_pipes.txt

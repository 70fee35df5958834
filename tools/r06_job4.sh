set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
for w in plonk marlin; do
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  TIMELINE_DUMP=$O/${w}_kernels.txt python tools/proof_timeline.py $DB 5 > $O/${w}_timeline.txt 2>&1
done

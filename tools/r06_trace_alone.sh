set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import bench; print(bench.host_demo_exe())"
O=gpurun_out/r06a; mkdir -p $O
for w in plonk marlin; do
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone.json 2>&1
  tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 > $O/${w}_4inflight.json 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  python tools/proof_timeline.py $DB 5 > $O/${w}_timeline.txt 2>&1
  python tools/timeline_gaps.py $DB > $O/${w}_gaps.txt 2>&1
done
cat $O/*.json

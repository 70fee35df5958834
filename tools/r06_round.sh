# round 6: the judged profiles of HEAD (run through gpurun from the repo root)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
O=gpurun_out/r06; mkdir -p $O
python -c "import bench; print(bench.host_demo_exe())"
for w in plonk marlin; do
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 --breakdown > $O/cpp_${w}_alone.json 2>&1
  tools/host_demo.bin $w --inflight 4 --steps 12 --warmup 2 > $O/cpp_${w}_4inflight.json 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  python tools/proof_timeline.py $DB 5 > $O/timeline_${w}_alone.txt 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof4_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 4 --steps 12 --warmup 2 > /tmp/prof4_$w.log 2>&1)
  DB=$(find /tmp/prof4_$w -name '*.db' | head -1)
  python tools/timeline_gaps.py $DB > $O/timeline_gaps_${w}_4inflight.txt 2>&1
done
ls -la $O | head -40
tail -5 gpurun_out/r06_profile_round.log

# round 6: the marks API, the hoisted Plonk / Marlin schedules from the compiled host, power sysfs probe
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
ls /sys/class/drm/ > $O/sysfs.txt; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo "$f $(cat $f 2>&1)"; done >> $O/sysfs.txt 2>&1
timeout 600 python -m pytest tests/test_marks.py -m gpu -x -q > $O/test_marks.txt 2>&1; tail -3 $O/test_marks.txt
for w in plonk marlin; do
  tools/host_demo.bin $w --inflight 1 --steps 4 --warmup 2 > $O/${w}_alone.json 2>&1
  tools/host_demo.bin $w --inflight 4 --steps 8 --warmup 2 > $O/${w}_4inflight.json 2>&1
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_$w -o $w -- $GRAFT_REPO_ROOT/tools/host_demo.bin $w --inflight 1 --steps 2 --warmup 1 > /tmp/prof_$w.log 2>&1)
  DB=$(find /tmp/prof_$w -name '*.db' | head -1)
  python tools/proof_timeline.py $DB 5 > $O/${w}_timeline.txt 2>&1
done
cat $O/*.json
timeout 900 python -m pytest tests/test_pipelines.py -m gpu -x -q > $O/test_pipelines.txt 2>&1; tail -5 $O/test_pipelines.txt

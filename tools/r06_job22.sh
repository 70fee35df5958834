cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06v; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace -d /tmp/tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --log-n 22 --steps 5 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report > $O/bench.json 2>/dev/null
cp $(find /tmp/tr -name '*.db' | head -1) $O/trace.db
cd $GRAFT_REPO_ROOT && python tools/timeline_gaps.py $O/trace.db | tee $O/gaps.txt

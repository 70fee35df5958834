cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06r; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace -d /tmp/tr -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report > $O/bench.json 2>/dev/null
cp $(find /tmp/tr -name '*.db' | head -1) $O/trace.db
ls -la $O

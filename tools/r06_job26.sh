cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06z; mkdir -p $O; rm -f $O/*
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python -m pytest tests/test_chaos.py -m gpu -x -q --durations=3 > $O/run_$i.txt 2>&1
  grep -E "passed|failed|call " $O/run_$i.txt | tee -a $O/chaos_repeat.txt
done

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06s; mkdir -p $O; rm -f $O/*
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q > $O/suite_$i.txt 2>&1; tail -4 $O/suite_$i.txt
done

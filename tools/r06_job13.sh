cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06m; mkdir -p $O
for rep in 1 2; do for g in 64 62 61; do python tools/stage_bench.py 20 4 msm_lane_interleave=$g 2>/dev/null | tee -a $O/stage.txt; done; done

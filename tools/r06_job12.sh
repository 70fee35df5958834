cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06l; mkdir -p $O
for rep in 1 2; do for g in 1 4 64; do python tools/stage_bench.py 20 4 msm_lane_interleave=$g 2>/dev/null | tee -a $O/stage.txt; done; done
for g in 1 3 64; do python tools/stage_bench.py 20 3 msm_lane_interleave=$g 2>/dev/null | tee -a $O/stage.txt; done
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
for g in 4 64 4 64; do python bench.py $C --ctx-option msm_lane_interleave=$g 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('il', $g, round(j['value'], 3), round(j['ms_per_step'], 3), j['results_checked'], round(j['roofline']['avg_launch_ms'], 3), round(j['roofline_g2']['avg_launch_ms'], 3), j['config']['results_sha256'][:12])" | tee -a $O/bench.txt; done

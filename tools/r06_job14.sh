cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06n; mkdir -p $O
AB=$PWD/collaborative-zksnark_amd/libczk_hip_ntt4.so
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
for rep in 1 2; do
  python bench.py $C 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base ', round(j['value'], 3), round(j['ms_per_step'], 3), j['results_checked'], 'busy', round(j['accumulate_busy_frac'], 3), j['stream_elapsed_ms_per_step']['ntt_pass'], j['config']['results_sha256'][:12])" | tee -a $O/bench.txt
  CZK_LIB_PATH=$AB python bench.py $C 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntt4 ', round(j['value'], 3), round(j['ms_per_step'], 3), j['results_checked'], 'busy', round(j['accumulate_busy_frac'], 3), j['stream_elapsed_ms_per_step']['ntt_pass'], j['config']['results_sha256'][:12])" | tee -a $O/bench.txt
done
python tools/ntt_bench.py 2>/dev/null | tail -6 | tee -a $O/ntt.txt
CZK_LIB_PATH=$AB python tools/ntt_bench.py 2>/dev/null | tail -6 | tee -a $O/ntt.txt

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06p; mkdir -p $O
LAB=$PWD/collaborative-zksnark_amd/libczk_hip_lab.so
C="--steps 20 --warmup 3 --no-cpu-baseline --no-seam-report --no-other-workloads --no-verify-report"
P='import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(j["value"], 3), round(j["ms_per_step"], 3), j["results_checked"], "busy", round(j["accumulate_busy_frac"], 3), {k: round(v, 1) for k, v in j["stream_elapsed_ms_per_step"].items() if k != "note"})'
for rep in 1 2; do
  CZK_LIB_PATH=$LAB python bench.py $C 2>/dev/null | python -c "$P" lab_base | tee -a $O/bench.txt
  CZK_LIB_PATH=$LAB CZK_NTT_SKIP_COSET_FIRST=1 python bench.py $C --no-result-check 2>/dev/null | python -c "$P" lab_nttskip | tee -a $O/bench.txt
done

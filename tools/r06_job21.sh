cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06u; mkdir -p $O
timeout 900 python -m pytest tests/test_verify.py -m gpu -x -q -k "marlin" --durations=3 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python tools/cpu_baseline_validate.py gpurun_out/r06_cpu_baseline_validation.json 14 20 2>&1 | tail -4

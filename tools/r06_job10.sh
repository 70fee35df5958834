cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
timeout 1500 python -m pytest tests/test_net.py tests/test_opens.py tests/test_pipelines.py tests/test_verify.py tests/test_rust_shim.py tests/test_reference_constants.py -m gpu -x -q --durations=8 > $O/gpu_tests_rest.txt 2>&1; tail -16 $O/gpu_tests_rest.txt

#!/bin/bash
# randomised differential soak on the GPU box: tests/test_gpu_parity.py::test_soak_with_a_fresh_seed for $1 seconds (default 600) on a fresh seed
# (or CZK_SOAK_SEED); the log goes to gpurun_out/soak/ and, when it is worth keeping, to profiles/.
set -u
SECS=${1:-600}
OUT=$PWD/gpurun_out/soak; mkdir -p $OUT
CZK_SOAK_SECONDS=$SECS timeout $((SECS + 600)) python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k soak_with_a_fresh_seed 2>&1 | tee $OUT/soak.txt | tail -12

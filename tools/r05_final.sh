#!/bin/bash
# final round-5 pass on the GPU box: the changed tests, the default bench line, then tools/profile_round.sh (kernel trace + PMC passes at the final sources)
set -u
R=$PWD; OUT=$R/gpurun_out/r05f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_net.py tests/test_pipelines.py -q -m gpu -x -k "party" --durations=6 2>&1 | tail -14
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json
bash tools/profile_round.sh r05f > $OUT/profile_round.log 2>&1; tail -4 $OUT/profile_round.log

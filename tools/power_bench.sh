#!/bin/bash
# Board power beside sustained loops of each candidate multiply instruction (tools/power_bench.hip); prints rate and the power samples.
cd "$(dirname "$0")"
for m in 0 1 2 3 4 5; do
  ./power_bench.bin $m 5 > /tmp/pb_$m.txt &
  pid=$!
  sleep 1.5
  samples=""
  for k in 1 2 3 4 5; do
    w=$(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$' | head -1)
    samples="$samples $w"
    sleep 0.3
  done
  wait $pid
  echo "$(cat /tmp/pb_$m.txt)   power samples (W):$samples"
done
echo "idle:$(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$' | head -1) W"
# the real workload: pipelined Groth16 proofs (accumulate kernels back to back); samples taken during the whole run, the timed steps are the plateau
python ../bench.py --steps 80 --warmup 2 --no-cpu-baseline --no-seam-report --no-result-check > /tmp/pb_bench.json 2>/dev/null &
pid=$!
samples=""
while kill -0 $pid 2>/dev/null; do
  w=$(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$' | head -1)
  samples="$samples $w"
  sleep 0.4
done
echo "bench.py --steps 80 (Groth16 2^20, pipelined): power samples over the run (W):$samples"
python3 -c "import json; d=json.loads(open('/tmp/pb_bench.json').read().strip().splitlines()[-1]); print('  ->', round(d['value'],2), 'proofs/s', round(d['ms_per_step'],2), 'ms per proof')"

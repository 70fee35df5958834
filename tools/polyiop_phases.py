#!/usr/bin/env python3
"""Per-round wall time of ONE Plonk / Marlin proof in flight: the time between consecutive transcript points (each is a full drain of the
context), with the number of MSM lanes, NTT lanes and library calls issued in between, and the host time spent enqueueing.
    python tools/polyiop_phases.py plonk|marlin [log_n] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import czk_amd as czk  # noqa: E402
from czk_amd import polyvm  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "plonk"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else (18 if what == "plonk" else 20)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = 1 << log_n
ctx = polyvm.shared_stream_context(czk, 0)
if what == "plonk":
    B = polyvm.GpuBackend(czk, ctx, 3, polyvm.plonk_max_degree(n))
    inp, prove = polyvm.plonk_inputs(B, n), polyvm.plonk_prove
else:
    B = polyvm.GpuBackend(czk, ctx, 4, polyvm.marlin_max_degree(n), lift=(1, 1, 0, 0))
    inp, prove = polyvm.marlin_inputs(B, n), polyvm.marlin_prove
ctx.sync()
prove(B, inp)
prove(B, inp)
marks = []
orig = B.transcript_point


def tp():
    t_enq = time.perf_counter()
    orig()
    marks.append((time.perf_counter(), t_enq, B.msm_count, B.ntt_count, B.msm_points))


B.transcript_point = tp
rows = None
for _ in range(reps):
    marks.clear()
    B.msm_count = B.ntt_count = B.msm_points = 0
    ctx.sync()
    t0 = time.perf_counter()
    prove(B, inp)
    ctx.sync()
    t1 = time.perf_counter()
    cur, prev, pm, pn, pp = [], t0, 0, 0, 0
    for t, t_enq, m, nn, pts in marks:
        cur.append(((t - prev) * 1e3, (t_enq - prev) * 1e3, m - pm, nn - pn, pts - pp))
        prev, pm, pn, pp = t, m, nn, pts
    cur.append(((t1 - prev) * 1e3, 0.0, 0, 0, 0))
    rows = cur if rows is None else [tuple(a + b for a, b in zip(r, c)) for r, c in zip(rows, cur)]
print(f"{what} 2^{log_n}: one proof in flight, {reps} proofs averaged")
tot = 0
for i, r in enumerate(rows):
    ms, enq, m, nn, pts = (v / reps for v in r)
    tot += ms
    acc = pts * 13 / 6.5e6     # ms of accumulation at the kernel's isolated rate (13 windows, 6.5 G additions/s)
    print(f"  phase {i}: {ms:7.2f} ms   host enqueue {enq:6.2f} ms   msm lanes {m:4.0f} ({pts / 1e6:6.2f} M point-lanes ~ {acc:5.1f} ms accumulate)   ntt lanes {nn:4.0f}")
print(f"  total {tot:.1f} ms")

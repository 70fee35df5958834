#!/usr/bin/env python3
"""Isolated (non-overlapped) stage times of one blocking MSM: sort / accumulate / reduce spans from the library's
HIP-event profiling hooks.  python tools/stage_bench.py [log_n] [lanes] [NAME=VALUE context options ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import czk_amd as czk
from util import rand_fr_canonical

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:]}
ctx = czk.Context(0, ts.cuda_stream, options=opts)
n = 1 << log_n
for g in (czk.CZK_G1, czk.CZK_G2):
    aw = 12 if g == czk.CZK_G1 else 24
    k = torch.from_numpy(rand_fr_canonical(0xBA5E5, n).view(np.int64)).cuda()
    pts = torch.empty((n, aw), dtype=torch.int64, device="cuda")
    ctx.fixed_base_points(g, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
    b = ctx.register_bases(g, pts.data_ptr(), None, n=n, mem=czk.CZK_MEM_DEVICE)
    s = torch.from_numpy(rand_fr_canonical(0xC0FFEE, lanes * n).view(np.int64)).cuda()
    ctx.msm(b, s.data_ptr(), n_scalars=n, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 4
    for _ in range(reps):
        ctx.msm(b, s.data_ptr(), n_scalars=n, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
    out = {}
    for name in ("msm_sort", f"msm_accumulate_g{g}", "msm_reduce"):
        ms, launches = ctx.profile_read(name)
        out[name] = round(ms / max(launches, 1), 3)
    ctx.profile_enable(False)
    print(f"G{g} n=2^{log_n} lanes={lanes} {opts}: {out}")
    b.release()

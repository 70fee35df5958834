#!/usr/bin/env python3
"""Window width per call: an MSM of m scalars under a key of n bases, once on the key's own tables (option "msm_fixed_c") and once with
the secondary table sets (default).  Reports blocking and pipelined time per MSM and the stage spans.
python tools/width_bench.py [log_key=20.58 (6*2^18)] [lanes=3]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import czk_amd as czk  # noqa: E402
from util import rand_fr_canonical  # noqa: E402


def run(n_key, sizes, lanes, fixed):
    ctx = czk.Context(0, options={"msm_fixed_c": 1 if fixed else 0})
    k = torch.from_numpy(rand_fr_canonical(5, n_key).view(np.int64)).cuda()
    pts = torch.empty((n_key, 12), dtype=torch.int64, device="cuda")
    ctx.fixed_base_points(1, k.data_ptr(), out=pts.data_ptr(), n=n_key, mem=czk.CZK_MEM_DEVICE)
    b = ctx.register_bases(1, pts.data_ptr(), None, n=n_key, mem=czk.CZK_MEM_DEVICE)
    print(f"key n={n_key} layout={b.layout()} fixed_c={fixed}")
    for m in sizes:
        s = torch.from_numpy(rand_fr_canonical(6, lanes * m).view(np.int64)).cuda()
        reps = 20
        outs = [np.zeros((lanes, 18), dtype=np.uint64) for _ in range(reps)]
        ctx.msm(b, s.data_ptr(), n_scalars=m, lanes=lanes, mem=czk.CZK_MEM_DEVICE)      # builds the set / sizes the workspaces
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(4):
            ctx.msm(b, s.data_ptr(), n_scalars=m, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
        blk = (time.perf_counter() - t0) / 4 * 1e3
        st = {nm: ctx.profile_read(nm) for nm in ("msm_sort", "msm_accumulate_g1", "msm_reduce")}
        ctx.profile_reset()
        t0 = time.perf_counter()
        for o in outs:
            ctx.msm_async(b, s.data_ptr(), m, lanes, czk.CZK_SCALAR_CANONICAL, o, stable=True)
        ctx.sync()
        pipe = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile_enable(False)
        print(f"  m={m:8d} lanes={lanes} layout_for={b.layout_for(m)}: blocking {blk:6.2f} ms  pipelined {pipe:6.2f} ms/MSM   isolated stages: "
              + ", ".join(f"{nm[4:]} {v[0] / max(1, v[1]):.2f}" for nm, v in st.items()))
    b.release()
    ctx.close()


if __name__ == "__main__":
    n_key = 6 * (1 << 18) + 1
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sizes = [1 << 12, 1 << 15, 1 << 17, (1 << 18) + 5, 3 * (1 << 17), 1 << 19, 3 * (1 << 18), n_key]
    run(n_key, sizes, lanes, True)
    run(n_key, sizes, lanes, False)

#!/usr/bin/env python3
"""How much do the sort / reduce stages of neighbouring MSMs cost the accumulate stream?  Issues the same 4-lane 2^20-point G1
MSM `reps` times back to back through czk_msm_async (stages overlap on the library's three streams) and compares the time per
MSM with the isolated accumulate kernel (tools/stage_bench.py).  Run on the GPU box: python tools/pipe_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import czk_amd as czk  # noqa: E402
from util import rand_fr_canonical  # noqa: E402


def main():
    ctx = czk.Context(0)
    lanes, reps = 4, 40
    for g, n in ((1, (1 << 20) + 1), (2, (1 << 20) + 1)):
        k = torch.from_numpy(rand_fr_canonical(5, n).view(np.int64)).cuda()
        pts = torch.empty((n, 12 * g), dtype=torch.int64, device="cuda")
        ctx.fixed_base_points(g, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
        b = ctx.register_bases(g, pts.data_ptr(), None, n=n, mem=czk.CZK_MEM_DEVICE)
        s = torch.from_numpy(rand_fr_canonical(6, lanes * n).view(np.int64)).cuda()
        outs = [np.zeros((lanes, 18 * g), dtype=np.uint64) for _ in range(reps)]
        torch.cuda.synchronize()
        for o in outs[:4]:
            ctx.msm_async(b, s.data_ptr(), n, lanes, czk.CZK_SCALAR_CANONICAL, o, stable=True)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        for o in outs:
            ctx.msm_async(b, s.data_ptr(), n, lanes, czk.CZK_SCALAR_CANONICAL, o, stable=True)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile_enable(False)
        acc = ctx.profile_read("msm_accumulate_g%d" % g)
        srt, red = ctx.profile_read("msm_sort"), ctx.profile_read("msm_reduce")
        print(f"G{g} 2^20+1 x {lanes} lanes, {reps} MSMs pipelined: {dt:.2f} ms per MSM; accumulate kernel {acc[0] / acc[1]:.2f} ms, sort stage {srt[0] / srt[1]:.2f} ms, "
              f"reduce stage {red[0] / red[1]:.2f} ms (stream times under overlap)")
        b.release()


if __name__ == "__main__":
    main()

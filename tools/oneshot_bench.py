#!/usr/bin/env python3
"""Times the one-shot entry points czk_msm_g1 / czk_msm_g2 (host buffers, bases not kept -- the reference's own argument
order) against registering the same bases with window tables first.  Run on the GPU box: python tools/oneshot_bench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import czk_amd as czk  # noqa: E402
from util import rand_fr_canonical  # noqa: E402


def main():
    ctx = czk.Context(0)
    for g, log_n in ((1, 16), (1, 20), (2, 18)):
        n = 1 << log_n
        k = rand_fr_canonical(7 + g, n)
        pts = ctx.fixed_base_points(g, k)
        s = rand_fr_canonical(99, n)
        ctx.msm_oneshot(g, pts[:64], None, s[:64])                       # warm-up (workspaces, streams)
        t0 = time.perf_counter()
        a = ctx.msm_oneshot(g, pts, None, s)
        t_one = time.perf_counter() - t0
        t0 = time.perf_counter()
        b = ctx.register_bases(g, pts, None)
        t_reg = time.perf_counter() - t0
        t0 = time.perf_counter()
        c = ctx.msm(b, s)
        t_msm = time.perf_counter() - t0
        same = np.array_equal(ctx.jac_to_affine(g, a)[0], ctx.jac_to_affine(g, c)[0])
        print(f"G{g} n=2^{log_n}: one-shot (no tables) {t_one * 1e3:8.1f} ms | register with tables {t_reg * 1e3:8.1f} ms + msm {t_msm * 1e3:6.1f} ms | same result: {same}")
        b.release()


if __name__ == "__main__":
    main()

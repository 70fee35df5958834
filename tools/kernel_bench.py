#!/usr/bin/env python3
"""Per-kernel timings mirroring the reference's bench shapes (algebra/poly-benches/benches/fft.rs:13-15: FFT
2^15..2^22; curves/curve-benches/src/macros/ec.rs:199-213: msm_131072) plus the BASELINE sizes.  Device-resident
inputs, HIP-event timing through the library's profiling hooks.  Prints a markdown table and writes JSON.

    python tools/kernel_bench.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import czk_amd as czk
from util import rand_fr_canonical

ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
ctx = czk.Context(0, ts.cuda_stream)
rows = []


def timed(fn, reps):
    fn()
    ctx.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("| kernel | size | lanes | ms | algorithmic GB/s | frac of 8 TB/s | field mul/s |")
print("|---|---|---|---|---|---|---|")
for log_d in (15, 17, 19, 20, 21, 22, 23):
    for lanes in (1, 4):
        d = 1 << log_d
        x = torch.from_numpy(rand_fr_canonical(log_d, 4096).view(np.int64)).cuda().repeat((lanes * d) // 4096 + 1, 1)[: lanes * d].contiguous()
        dt = timed(lambda: ctx.ntt_fr(x.data_ptr(), log_d, czk.CZK_FFT, lanes=lanes, mem=czk.CZK_MEM_DEVICE), 5)
        gb = lanes * 2 * d * 32 / dt / 1e9                      # 2 * D * 32 B per lane
        muls = lanes * (d // 2) * log_d / dt
        rows.append({"kernel": "ntt_fft", "log_size": log_d, "lanes": lanes, "ms": dt * 1e3, "alg_gbs": gb, "frac": gb / 8000, "mul_per_s": muls})
        print(f"| NTT (fft_in_place) | 2^{log_d} | {lanes} | {dt*1e3:.3f} | {gb:.0f} | {gb/8000:.3f} | {muls:.3g} Fr |")
        del x
for g, sizes in ((czk.CZK_G1, (17, 20, 21, 22)), (czk.CZK_G2, (17, 20))):
    for log_n in sizes:
        n = 1 << log_n
        aw = 12 if g == czk.CZK_G1 else 24
        k = torch.from_numpy(rand_fr_canonical(0xBA5E5, n).view(np.int64)).cuda()
        pts = torch.empty((n, aw), dtype=torch.int64, device="cuda")
        ctx.fixed_base_points(g, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
        b = ctx.register_bases(g, pts.data_ptr(), None, n=n, mem=czk.CZK_MEM_DEVICE)
        del pts, k
        for lanes in (1, 4):
            s = torch.from_numpy(rand_fr_canonical(0xC0FFEE + lanes, lanes * n).view(np.int64)).cuda()
            dt = timed(lambda: ctx.msm(b, s.data_ptr(), n_scalars=n, lanes=lanes, mem=czk.CZK_MEM_DEVICE), 3)
            gb = (n * aw * 8 + lanes * n * 32) / dt / 1e9        # bases once + scalars per lane
            rows.append({"kernel": f"msm_g{g}", "log_size": log_n, "lanes": lanes, "ms": dt * 1e3, "alg_gbs": gb, "frac": gb / 8000,
                         "points_per_s": lanes * n / dt})
            print(f"| MSM G{g} (blocking czk_msm) | 2^{log_n} | {lanes} | {dt*1e3:.2f} | {gb:.1f} | {gb/8000:.5f} | {lanes*n/dt:.3g} pts/s |")
            del s
        b.release()
# ---- callers either side of the NTT: constraint evaluation and division by (X - z) ------------------------------------
for log_m in (20, 22):
    m, lanes = 1 << log_m, 4
    one = np.array([9015221291577245683, 8239323489949974514, 1646089257421115374, 958099254763297437], dtype=np.uint64)   # R mod r
    for kind in ("unit", "dense3"):
        if kind == "unit":            # the squaring circuit's shape: one term per row, coefficient 1
            rp = np.arange(m + 1, dtype=np.uint64)
            col = np.arange(m, dtype=np.uint32)
            cf = np.tile(one, (m, 1))
        else:                          # 3 terms per row, general coefficients, random columns
            rp = (3 * np.arange(m + 1)).astype(np.uint64)
            col = np.random.default_rng(1).integers(0, m, size=3 * m).astype(np.uint32)
            cf = rand_fr_canonical(5, 4096)[np.arange(3 * m) % 4096]
        mat = ctx.r1cs_matrix_register(rp, col, cf, m)
        z = torch.from_numpy(rand_fr_canonical(7, 4096).view(np.int64)).cuda().repeat(lanes * m // 4096, 1).contiguous()
        out = torch.empty((lanes, m, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        dt = timed(lambda: ctx.r1cs_matvec(mat, z.data_ptr(), lanes=lanes, out=out.data_ptr(), z_stride=m, out_stride=m, mem=czk.CZK_MEM_DEVICE), 5)
        nnz = col.size
        alg = m * 4 + nnz * 4 + (0 if kind == "unit" else nnz * 32) + lanes * (nnz * 32 + m * 32)   # row_ptr + col (+ coeff) + gathers + outputs
        rows.append({"kernel": f"r1cs_matvec_{kind}", "log_size": log_m, "lanes": lanes, "ms": dt * 1e3, "alg_gbs": alg / dt / 1e9, "frac": alg / dt / 8e12})
        print(f"| R1CS mat-vec ({kind}) | 2^{log_m} rows | {lanes} | {dt*1e3:.3f} | {alg/dt/1e9:.0f} | {alg/dt/8e12:.3f} | |")
        mat.release()
        del z, out
for log_n in (20, 21, 23):
    n, lanes = 1 << log_n, 2
    p = torch.from_numpy(rand_fr_canonical(9, 4096).view(np.int64)).cuda().repeat(lanes * n // 4096, 1).contiguous()
    q = torch.empty((lanes, n - 1, 4), dtype=torch.int64, device="cuda")
    r = torch.empty((lanes, 4), dtype=torch.int64, device="cuda")
    zpt = rand_fr_canonical(10, 1)[0]
    torch.cuda.synchronize()
    dt = timed(lambda: ctx.poly_div_linear(p.data_ptr(), zpt, lanes=lanes, n=n, quotient=q.data_ptr(), remainder=r.data_ptr(), mem=czk.CZK_MEM_DEVICE), 5)
    alg = lanes * 2 * n * 32                                      # read the coefficients once, write the quotient once
    rows.append({"kernel": "poly_div_linear", "log_size": log_n, "lanes": lanes, "ms": dt * 1e3, "alg_gbs": alg / dt / 1e9, "frac": alg / dt / 8e12,
                 "mul_per_s": lanes * 2 * n / dt})
    print(f"| poly / (X - z) | 2^{log_n} | {lanes} | {dt*1e3:.3f} | {alg/dt/1e9:.0f} | {alg/dt/8e12:.3f} | {lanes*2*n/dt:.3g} Fr |")
    del p, q, r
for log_n in (20, 22):
    n = 1 << log_n
    x = torch.from_numpy(rand_fr_canonical(11, 4096).view(np.int64)).cuda().repeat(n // 4096, 1).contiguous()
    x = torch.from_numpy(np.ascontiguousarray(x.cpu().numpy())).cuda()
    o = torch.empty_like(x)
    torch.cuda.synchronize()
    xm = torch.empty_like(x)
    ctx.fr_from_repr(x.data_ptr(), out=xm.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
    ctx.sync()
    for name, fn, muls in (("fr_prefix_product", lambda: ctx.fr_prefix_product(xm.data_ptr(), out=o.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE), 2),
                           ("fr_batch_inverse", lambda: ctx.fr_batch_inverse(xm.data_ptr(), out=o.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE), 3 + 380 / 64)):
        dt = timed(fn, 5)
        alg = 2 * n * 32
        rows.append({"kernel": name, "log_size": log_n, "lanes": 1, "ms": dt * 1e3, "alg_gbs": alg / dt / 1e9, "frac": alg / dt / 8e12, "mul_per_s": muls * n / dt})
        print(f"| {name} | 2^{log_n} | 1 | {dt*1e3:.3f} | {alg/dt/1e9:.0f} | {alg/dt/8e12:.3f} | {muls*n/dt:.3g} Fr |")
    del x, o, xm
for log_n in (21,):   # pointwise steps of the witness map
    n, lanes = 1 << log_n, 4
    a = torch.from_numpy(rand_fr_canonical(12, 4096).view(np.int64)).cuda().repeat(lanes * n // 4096, 1).contiguous()
    b = a.clone()
    o = torch.empty_like(a)
    torch.cuda.synchronize()
    for name, fn, nbuf in (("fr_vec_op(MUL)", lambda: ctx.fr_vec_op(2, a.data_ptr(), b.data_ptr(), out=o.data_ptr(), n=lanes * n, mem=czk.CZK_MEM_DEVICE), 3),
                           ("fr_vec_op(ADD)", lambda: ctx.fr_vec_op(0, a.data_ptr(), b.data_ptr(), out=o.data_ptr(), n=lanes * n, mem=czk.CZK_MEM_DEVICE), 3)):
        dt = timed(fn, 10)
        alg = nbuf * lanes * n * 32
        rows.append({"kernel": name, "log_size": log_n, "lanes": lanes, "ms": dt * 1e3, "alg_gbs": alg / dt / 1e9, "frac": alg / dt / 8e12})
        print(f"| {name} | 2^{log_n} | {lanes} | {dt*1e3:.3f} | {alg/dt/1e9:.0f} | {alg/dt/8e12:.3f} | |")
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)

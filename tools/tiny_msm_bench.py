"""Latency of short MSMs (KZG hiding commitments multiply 2 - 3 points; poly-commit/src/kzg10/mod.rs:165-192): blocking and pipelined time per call,
4 share lanes, keys registered at the call size.  Run on the GPU box: python tools/tiny_msm_bench.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import importlib
czk = importlib.import_module("collaborative-zksnark_amd.binding")
from util import rand_fr_canonical
ctx = czk.Context()
for n in (2, 3, 16, 64, 256, 1024, 4096):
    k = rand_fr_canonical(n, n)
    b = ctx.register_bases(1, ctx.fixed_base_points(1, k), None)
    lanes = 4
    s = rand_fr_canonical(7, lanes * n).reshape(lanes, n, 4)
    sd = torch.from_numpy(s.view(np.int64)).cuda(); torch.cuda.synchronize()
    lay = b.layout() if hasattr(b, "layout") else None
    for _ in range(3): ctx.msm(b, sd.data_ptr(), n_scalars=n, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
    t0 = time.perf_counter()
    for _ in range(30): ctx.msm(b, sd.data_ptr(), n_scalars=n, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
    tb = (time.perf_counter() - t0) / 30
    outs = [np.zeros((lanes, 18), dtype=np.uint64) for _ in range(30)]
    t0 = time.perf_counter()
    for o in outs: ctx.msm_async(b, sd.data_ptr(), n, lanes, czk.CZK_SCALAR_CANONICAL, o, stable=True)
    ctx.sync()
    tp = (time.perf_counter() - t0) / 30
    print(f"n={n} layout={lay} blocking {tb*1e3:.3f} ms pipelined {tp*1e3:.3f} ms", flush=True)

import sys, os, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, torch
import czk_amd as czk
from util import rand_fr_canonical
ctx = czk.Context(0)
for log_n in (20, 21, 23):
    n, lanes = 1 << log_n, 2
    p = torch.from_numpy(rand_fr_canonical(9, 4096).view(np.int64)).cuda().repeat(lanes * n // 4096, 1).contiguous()
    q = torch.empty((lanes, n - 1, 4), dtype=torch.int64, device="cuda")
    r = torch.empty((lanes, 4), dtype=torch.int64, device="cuda")
    z = rand_fr_canonical(10, 1)[0]
    torch.cuda.synchronize()
    f = lambda: ctx.poly_div_linear(p.data_ptr(), z, lanes=lanes, n=n, quotient=q.data_ptr(), remainder=r.data_ptr(), mem=czk.CZK_MEM_DEVICE)
    f(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10): f()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 10
    print(log_n, round(dt * 1e3, 3), "ms", round(lanes * 2 * n * 32 / dt / 1e9), "GB/s")

import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np, torch
import czk_amd as czk
from util import rand_fr_canonical
ctx = czk.Context(0)
for log_n in (16, 20):
    n = 1 << log_n
    k = torch.from_numpy(rand_fr_canonical(1, n).view(np.int64)).cuda()
    pts = torch.empty((n, 12), dtype=torch.int64, device="cuda")
    ctx.fixed_base_points(1, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
    b = ctx.register_bases(1, pts.data_ptr(), None, n=n, mem=czk.CZK_MEM_DEVICE)
    s = np.zeros((n, 4), dtype=np.uint64); s[:, 0] = 1
    s[::2] = rand_fr_canonical(2, n // 2)          # half boolean-one, half random
    sd = torch.from_numpy(s.view(np.int64)).cuda(); torch.cuda.synchronize()
    ctx.msm(b, sd.data_ptr(), n_scalars=n, lanes=1, mem=czk.CZK_MEM_DEVICE)
    t0 = time.perf_counter(); ctx.msm(b, sd.data_ptr(), n_scalars=n, lanes=1, mem=czk.CZK_MEM_DEVICE); dt = time.perf_counter() - t0
    print(f"n=2^{log_n}: half of the scalars equal to 1: {dt*1e3:.1f} ms")
    b.release()

"""What the host-memory seam (CZK_MEM_HOST) gets out of PCIe: czk_ntt_fr on pageable lanes against raw pinned / pageable copies of the same bytes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import czk_amd

def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

ctx = czk_amd.Context(0)
log_d, lanes = 21, 4
nbytes = lanes * (32 << log_d)
host = np.random.default_rng(1).integers(0, 1 << 60, size=(lanes, 1 << log_d, 4), dtype=np.uint64)
host[..., 3] &= (1 << 60) - 1
dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
pin = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
pag = torch.from_numpy(host.view(np.uint8).reshape(-1))
print(f"bytes {nbytes/2**20:.0f} MiB")
for name, src in (("pinned", pin), ("pageable", pag)):
    up = t(lambda: dev.copy_(src, non_blocking=True))
    dn = t(lambda: src.copy_(dev, non_blocking=True))
    print(f"raw {name}: H2D {nbytes/up/1e9:.1f} GB/s  D2H {nbytes/dn/1e9:.1f} GB/s")
# both directions at once, pinned, two streams
s2 = torch.cuda.Stream()
pin2 = torch.empty(nbytes, dtype=torch.uint8).pin_memory(); dev2 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
def duplex():
    dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2): pin2.copy_(dev2, non_blocking=True)
d = t(duplex)
print(f"raw pinned duplex: {nbytes/d/1e9:.1f} GB/s each way")
for kind in (czk_amd.CZK_IFFT, czk_amd.CZK_COSET_FFT):
    dt = t(lambda: ctx.ntt_fr(host, log_d, kind, lanes=lanes), reps=4)
    print(f"czk_ntt_fr host lanes kind {kind}: {dt*1e3:.1f} ms = {nbytes/dt/1e9:.1f} GB/s each way")
td = torch.from_numpy(host.view(np.int64)).cuda()
dt = t(lambda: (ctx.ntt_fr(td.data_ptr(), log_d, czk_amd.CZK_IFFT, lanes=lanes, mem=czk_amd.CZK_MEM_DEVICE), ctx.sync()))
print(f"czk_ntt_fr device: {dt*1e3:.2f} ms")

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, numpy as np
import czk_amd as czk
import bench
torch.cuda.set_device(0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
ctx = czk.Context(0, ts.cuda_stream)
p = bench.Groth16Local(czk, ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 2)
for it in range(4):
    p.step(); torch.cuda.synchronize()
    bad = p.chk.reshape(2, -1, 4).ne(0).any(dim=2)
    print("iter", it, "chk nonzero elems:", int(bad.sum().item()), "first idx:", bad.nonzero()[:3].tolist(), flush=True)
# serialized variant: sync after each enqueue
orig = ctx.msm_async
def ser(*a, **k):
    r = orig(*a, **k); ctx.sync(); return r
ctx.msm_async = ser
for it in range(2):
    p.step(); torch.cuda.synchronize()
    bad = p.chk.reshape(2, -1, 4).ne(0).any(dim=2)
    print("serialized iter", it, "chk nonzero elems:", int(bad.sum().item()), flush=True)

import os, sys, cProfile, pstats
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import czk_amd as czk
from czk_amd import polyvm
what = sys.argv[1]
n = 1 << (18 if what == "plonk" else 20)
ctx = polyvm.shared_stream_context(czk, 0)
if what == "plonk":
    B = polyvm.GpuBackend(czk, ctx, 3, polyvm.plonk_max_degree(n)); inp, prove = polyvm.plonk_inputs(B, n), polyvm.plonk_prove
else:
    B = polyvm.GpuBackend(czk, ctx, 4, polyvm.marlin_max_degree(n), lift=(1, 1, 0, 0)); inp, prove = polyvm.marlin_inputs(B, n), polyvm.marlin_prove
prove(B, inp); prove(B, inp); ctx.sync()
pr = cProfile.Profile()
pr.enable()
prove(B, inp); ctx.sync()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

#!/usr/bin/env python3
"""NTT timings, device-resident, all four kinds: `python tools/ntt_bench.py [log_d ...]` (CZK_NTT_GEN1=1 selects the
first-generation passes for every size)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import czk_amd as czk
from util import rand_fr_canonical

sizes = [int(a) for a in sys.argv[1:]] or [18, 20, 21, 22, 23]
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
ctx = czk.Context(0, ts.cuda_stream)
gen = "gen1" if os.environ.get("CZK_NTT_GEN1") else "gen2"
for log_d in sizes:
    for lanes in (1, 4):
        d = 1 << log_d
        x = torch.from_numpy(rand_fr_canonical(3, 4096).view(np.int64)).cuda().repeat(lanes * d // 4096, 1).contiguous()
        for kind, name in ((czk.CZK_FFT, "fft"), (czk.CZK_IFFT, "ifft"), (czk.CZK_COSET_FFT, "coset_fft"), (czk.CZK_COSET_IFFT, "coset_ifft")):
            for _ in range(2):
                ctx.ntt_fr(x.data_ptr(), log_d, kind, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
            ctx.sync()
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.ntt_fr(x.data_ptr(), log_d, kind, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
            ctx.sync()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{gen} {name:10s} 2^{log_d} x {lanes}: {dt*1e3:8.3f} ms  {lanes*2*d*32/dt/1e9:7.0f} GB/s algorithmic  {lanes*(d//2)*log_d/dt/1e9:6.1f} G butterflies/s", flush=True)
        del x

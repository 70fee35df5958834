#!/usr/bin/env python3
"""A few device-resident NTTs of the BASELINE shape (2^21, 4 lanes, all four kinds) for rocprofv3 --pmc passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import czk_amd as czk
from util import rand_fr_canonical

log_d, lanes = 21, 4
d = 1 << log_d
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
ctx = czk.Context(0, ts.cuda_stream)
x = torch.from_numpy(rand_fr_canonical(3, 4096).view(np.int64)).cuda().repeat(lanes * d // 4096, 1).contiguous()
torch.cuda.synchronize()
for _ in range(3):
    for kind in (czk.CZK_FFT, czk.CZK_IFFT, czk.CZK_COSET_FFT, czk.CZK_COSET_IFFT):
        ctx.ntt_fr(x.data_ptr(), log_d, kind, lanes=lanes, mem=czk.CZK_MEM_DEVICE)
ctx.sync()
print("ok")

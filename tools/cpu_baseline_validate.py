#!/usr/bin/env python3
"""Validates the scaling model behind bench.py's single-thread `cpu_baseline` (a 2^14 sample scaled by the reference algorithm's
multiplication count, bench._ref_work): times the checker's serial proof-local compute (4 share lanes: witness map + 5 MSMs each)
at several sizes on ONE thread and compares measured ratios with the model's.

    python tools/cpu_baseline_validate.py OUT.json 12 14 16 [18 [20]]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import orc
from util import rand_fr_canonical
import importlib.util
_spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)

out_path, sizes = sys.argv[1], [int(a) for a in sys.argv[2:]]
rows = []
for log_n in sizes:
    N = 1 << log_n
    log_d = (N + 1).bit_length()
    D = 1 << log_d
    b1, b2 = orc.chain_points(1, D), orc.chain_points(2, N + 1)
    x = orc.fr_from_repr(rand_fr_canonical(5, D))
    inf = np.zeros(D, dtype=np.uint8)
    t0 = time.perf_counter()
    for _ in range(4):
        a, b = orc.witness_map_pre(x, x, log_d)
        h = orc.witness_map_post(orc.fr_mul(a, b), x, log_d)
        orc.multi_scalar_mul(1, b1[:D - 1], inf, h)
        orc.multi_scalar_mul(1, b1[:N], inf, x[:N])
        orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1])
        orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1])
        orc.multi_scalar_mul(2, b2[:N + 1], inf, x[:N + 1])
    dt = time.perf_counter() - t0
    rows.append({"log_n": log_n, "seconds": dt, "model_work": bench._ref_work(N)})
    print(rows[-1], flush=True)
base = rows[0]
for r in rows:
    r["measured_ratio_to_first"] = r["seconds"] / base["seconds"]
    r["model_ratio_to_first"] = r["model_work"] / base["model_work"]
    r["model_error"] = r["model_ratio_to_first"] / r["measured_ratio_to_first"] - 1
ref14 = next((r for r in rows if r["log_n"] == 14), None)
res = {"host": os.uname().nodename, "cpu_count": os.cpu_count(), "threads": 1, "lanes": 4, "rows": rows,
       "note": "model_error = (time predicted from the first row by the model) / (measured time) - 1"}
if ref14:
    for r in rows:
        r["predicted_from_2^14_s"] = ref14["seconds"] * r["model_work"] / ref14["model_work"]
json.dump(res, open(out_path, "w"), indent=1)

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one results.db per pass) into per-kernel averages.

    python tools/pmc_summary.py OUT.json DB [DB ...]

Each DB comes from `rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline`
(counters in separate passes, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are reported in KB."""
import json
import sqlite3
import sys

out_path, dbs = sys.argv[1], sys.argv[2:]
res = {}
for path in dbs:
    db = sqlite3.connect(path)
    q = ("select counter_name, kernel_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
         "group by counter_name, kernel_name order by counter_name, 4 desc")
    for name, kernel, n, tot, avg, dur in db.execute(q):
        res.setdefault(name, []).append({"kernel": kernel[:96], "dispatches": n, "sum": tot, "avg_per_dispatch": avg, "avg_duration_ns": dur})
summary = {"counters": {k: v[:12] for k, v in res.items()}}


def pick(counter, prefix):
    for r in res.get(counter, []):
        if r["kernel"].startswith(prefix):
            return r
    return None


XCDS = 8   # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)
g1_prefix = "czk::k_accumulate_te(" if any(r["kernel"].startswith("czk::k_accumulate_te(") for r in res.get("SQ_INSTS_VALU", []) + res.get("FETCH_SIZE", [])) else "czk::k_accumulate_u("
summary["g1_kernel"] = g1_prefix
for tag, key, prefix in (("g1", "k_accumulate_u", g1_prefix), ("g2", "k_accumulate_u2", "czk::k_accumulate_u2(")):
    f, w = pick("FETCH_SIZE", prefix), pick("WRITE_SIZE", prefix)
    if f and w:
        summary[f"msm_accumulate_{tag}_fetch_bytes_per_launch"] = f["avg_per_dispatch"] * 1024
        summary[f"msm_accumulate_{tag}_write_bytes_per_launch"] = w["avg_per_dispatch"] * 1024
        summary[f"msm_accumulate_{tag}_bytes_per_launch"] = (f["avg_per_dispatch"] + w["avg_per_dispatch"]) * 1024
    v, g = pick("SQ_INSTS_VALU", prefix), pick("GRBM_GUI_ACTIVE", prefix)
    wc, wv = pick("SQ_WAVE_CYCLES", prefix), pick("SQ_WAVES", prefix)
    if v:
        summary[f"{key}_sq"] = {"SQ_INSTS_VALU_per_launch": v["avg_per_dispatch"], "avg_duration_ns": v["avg_duration_ns"],
                                "GRBM_GUI_ACTIVE": g and g["avg_per_dispatch"], "SQ_WAVE_CYCLES": wc and wc["avg_per_dispatch"],
                                "SQ_WAVES": wv and wv["avg_per_dispatch"]}
    if g:
        summary[f"{key}_effective_clock_ghz"] = g["avg_per_dispatch"] / XCDS / g["avg_duration_ns"]
import hashlib
import os
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "collaborative-zksnark_amd", "csrc")
_h = hashlib.sha256()
for _fn in sorted(os.listdir(_d)):
    if _fn.endswith((".hip", ".h", ".inc")) and _fn != "net.hip":     # as bench.py csrc_digest(): the communicator is host code
        _h.update(_fn.encode())
        _h.update(open(os.path.join(_d, _fn), "rb").read())
summary["csrc_sha256"] = _h.hexdigest()[:16]      # bench.py csrc_digest(): were the counters taken at the kernels a bench line ran?
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}, indent=1))

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


f, w = pick("FETCH_SIZE", "czk::k_accumulate_u("), pick("WRITE_SIZE", "czk::k_accumulate_u(")
if f and w:
    summary["msm_accumulate_g1_fetch_bytes_per_launch"] = f["avg_per_dispatch"] * 1024
    summary["msm_accumulate_g1_write_bytes_per_launch"] = w["avg_per_dispatch"] * 1024
    summary["msm_accumulate_g1_bytes_per_launch"] = (f["avg_per_dispatch"] + w["avg_per_dispatch"]) * 1024
v, g = pick("SQ_INSTS_VALU", "czk::k_accumulate_u("), pick("GRBM_GUI_ACTIVE", "czk::k_accumulate_u(")
wc, wv = pick("SQ_WAVE_CYCLES", "czk::k_accumulate_u("), pick("SQ_WAVES", "czk::k_accumulate_u(")
if v:
    summary["k_accumulate_u_sq"] = {"SQ_INSTS_VALU_per_launch": v["avg_per_dispatch"], "avg_duration_ns": v["avg_duration_ns"],
                                    "GRBM_GUI_ACTIVE": g and g["avg_per_dispatch"], "SQ_WAVE_CYCLES": wc and wc["avg_per_dispatch"],
                                    "SQ_WAVES": wv and wv["avg_per_dispatch"]}
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}, indent=1))

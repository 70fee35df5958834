#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 --pmc result databases (generic version of pmc_summary.py).

    python tools/pmc_kernels.py OUT.json DB [DB ...]"""
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
        res.setdefault(kernel[:120], {})[name] = {"dispatches": n, "avg_per_dispatch": avg, "avg_duration_ns": dur}
json.dump(res, open(out_path, "w"), indent=1)
for k, v in res.items():
    print(k[:100], {n: round(c["avg_per_dispatch"]) for n, c in v.items()})

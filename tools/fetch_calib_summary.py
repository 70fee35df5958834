#!/usr/bin/env python3
"""FETCH_SIZE calibration factors from one `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/fetch_calib.bin` pass.

    python tools/fetch_calib_summary.py RESULTS.db STDOUT_OF_THE_RUN.txt > calibration.json

factor = bytes the counter reports / bytes the kernel is known to have requested from beyond its caches (k_gather: random 192-byte entries of a
multi-GB table, so every entry misses L2 and the Infinity Cache almost surely; k_stream: one linear pass).  roofline.traffic of bench.py divides
the raw counter of the accumulate kernels by the k_gather factor."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
known = {}
for ln in open(sys.argv[2]):
    if ln.startswith("{"):
        j = json.loads(ln)
        known[j["kernel"]] = j
out = {"what": "rocprofv3 FETCH_SIZE (KB per dispatch x 1024) against a known byte count, tools/fetch_calib.hip"}
q = "select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = 'FETCH_SIZE' group by kernel_name"
for kernel, n, avg_kb, dur in db.execute(q):
    for name, j in known.items():
        if name in kernel:
            out[name] = {"dispatches": n, "fetch_size_bytes_per_dispatch": avg_kb * 1024, "known_bytes_per_dispatch": j["bytes"],
                         "factor": avg_kb * 1024 / j["bytes"], "avg_duration_ms": dur / 1e6, "table_bytes": j.get("table_bytes"),
                         "known_gb_per_s": j["bytes"] / (dur / 1e9) / 1e9}
print(json.dumps(out, indent=1))

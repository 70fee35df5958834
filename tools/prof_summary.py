"""Summarise a rocprofv3 results.db (kernel trace) into per-kernel totals.  usage: prof_summary.py DB [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("%-64s %6s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct"))
for r in rows[:40]:
    print("%-64s %6d %11.3f %9.3f %9.3f %9.3f %5.1f%%" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))

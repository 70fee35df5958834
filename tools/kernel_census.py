"""Per-kernel call counts and total / longest durations of one rocprofv3 --kernel-trace database: python tools/kernel_census.py DB"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, max(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total kernel ms", tot)
for r in rows[:45]:
    print(f"{r[0][:90]:90s} {r[1]:6d} {r[2]:9.2f} ms  max {r[3]:.3f}")

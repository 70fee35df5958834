import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by kernel_name order by 4 desc")
for k, n, v, d in db.execute(q):
    if d > 2e5:
        print(f"{k[:70]:70s} n={n:4d} dur={d/1e6:8.3f} ms  clock={v/8/d:6.3f} GHz")

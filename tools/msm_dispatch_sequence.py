"""Time-ordered dispatches of the MSM sort / accumulate kernels of one rocprofv3 --kernel-trace database (start offset, duration, grid, workgroup):
python tools/msm_dispatch_sequence.py DB"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
print(cols)
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
t0 = rows[0][1]
want = ("k_digits_part", "k_part_sort", "k_part_scatter", "k_count_starts", "k_accumulate_te")
for r in rows:
    nm = r[0]
    if any(w in nm for w in want):
        short = [w for w in want if w in nm][0]
        print(f"{(r[1]-t0)/1e6:10.2f} ms  {short:16s} {(r[2]-r[1])/1e6:8.3f} ms  grid {r[3]}x{r[4]} wg {r[5]}")

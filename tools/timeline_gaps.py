#!/usr/bin/env python3
"""How busy is the GPU, and is the dominant kernel always running?  From a rocprofv3 --kernel-trace results.db:

    python tools/timeline_gaps.py DB [PATTERN] [SKIP]

over the window from SKIP (default 0.4) of the way through the PATTERN (default k_accumulate) dispatches to the last one: fraction of
wall time with any kernel running, with a PATTERN kernel running, and the longest stretches without one."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_accumulate"
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
rows = list(db.execute("select name, start, end from kernels order by start"))
dom = [(s, e) for k, s, e in rows if pat in k and "heavy" not in k and "fix" not in k and "cleanup" not in k]
t0, t1 = dom[int(len(dom) * skip)][0], max(e for _, e in dom)


def union(iv):
    iv = sorted((max(s, t0), min(e, t1)) for s, e in iv if e > t0 and s < t1)
    tot, cur_s, cur_e, gaps = 0, None, None, []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
                gaps.append((s - cur_e, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot, gaps


wall = t1 - t0
busy, _ = union([(s, e) for _, s, e in rows])
dbusy, gaps = union(dom)
print(f"window {wall / 1e6:.1f} ms, {sum(1 for s, e in dom if s >= t0)} '{pat}' dispatches")
print(f"any kernel running      {busy / wall:.3f} of the window")
print(f"'{pat}' kernel running  {dbusy / wall:.3f} of the window")
gaps.sort(reverse=True)
print("longest stretches without one (ms):", " ".join(f"{g / 1e6:.2f}" for g, _ in gaps[:12]), f"... {len(gaps)} gaps, {sum(g for g, _ in gaps) / 1e6:.1f} ms in all")

# what runs while no PATTERN kernel does: kernel time inside the gaps, by kernel name (overlapping kernels each count their own time)
gap_iv = sorted((at, at + g) for g, at in gaps)
import bisect
starts = [a for a, _ in gap_iv]
by_name = {}
for k, s, e in rows:
    if e <= t0 or s >= t1 or (pat in k and "heavy" not in k and "fix" not in k and "cleanup" not in k):
        continue
    i = max(0, bisect.bisect_right(starts, s) - 1)
    inside = 0
    while i < len(gap_iv) and gap_iv[i][0] < e:
        inside += max(0, min(e, gap_iv[i][1]) - max(s, gap_iv[i][0]))
        i += 1
    if inside:
        name = k.split("(")[0].replace("void ", "").replace("czk::", "")[:60]
        d = by_name.setdefault(name, [0, 0])
        d[0] += inside
        d[1] += 1
tot_gap = sum(g for g, _ in gaps)
print(f"kernel time inside the {tot_gap / 1e6:.1f} ms of gaps (a kernel spanning a gap counts the part inside; concurrent kernels add up):")
for name, (ns, cnt) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {ns / 1e6:8.2f} ms  {cnt:6d} x  {name}")
# idle time inside the gaps: no kernel at all
idle = 0
all_iv = sorted((max(s, t0), min(e, t1)) for _, s, e in rows if e > t0 and s < t1)
cur = t0
for s, e in all_iv:
    if s > cur:
        idle += s - cur
    cur = max(cur, e)
print(f"no kernel at all: {idle / 1e6:.2f} ms of the window")

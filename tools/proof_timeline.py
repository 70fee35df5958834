#!/usr/bin/env python3
"""One proof's timeline from a rocprofv3 --kernel-trace results.db of a run with ONE proof in flight:

    python tools/proof_timeline.py DB PROOFS [PATTERN] [WHICH]

The trace holds PROOFS proofs made one after another (warm-up included); their boundaries are the longest stretches with no kernel at
all (the host's drain between two proofs).  Prints, for proof WHICH (default: the last), the alternation of stretches in which a PATTERN
(default k_accumulate) kernel is running and the gaps between them, with the kernels that run inside every gap -- the dependency-forced
part of a single proof's wall time sits in those gaps."""
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
proofs = int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else "k_accumulate"
which = int(sys.argv[4]) if len(sys.argv) > 4 else proofs - 1
rows = list(db.execute("select name, start, end from kernels order by start"))


def is_dom(k):
    return pat in k and "heavy" not in k and "fix" not in k and "cleanup" not in k


dom = [(s, e) for k, s, e in rows if is_dom(k)]
per = len(dom) // proofs
# (set-up work before the first proof -- the index commitments of Marlin -- may add PATTERN kernels at the head of the trace: count from the tail)
# Every proof after the first enqueues the same kernels in the same order, and the trace ends with the last proof's last kernel: the number of kernels
# between the first PATTERN kernel of one proof and the first of the next is a proof's kernel count K, and proof `which` is K kernels ending where
# the following proofs' kernels begin.  (Idle stretches do not mark the boundary any more: the provers enqueue a proof's first commitment at once.)
by_start = sorted(range(len(rows)), key=lambda i: rows[i][1])
dom_pos = [i for i in by_start if is_dom(rows[i][0])]          # positions (in start order) of the PATTERN kernels
order = {i: n for n, i in enumerate(by_start)}
K = order[dom_pos[-1]] - order[dom_pos[-1 - per]]
hi = len(rows) - K * (proofs - 1 - which)
sel = [rows[i] for i in by_start[hi - K:hi]]
w0, w1 = min(s for _, s, _ in sel), max(e for _, _, e in sel)
name = lambda k: k.split("(")[0].replace("void ", "").replace("czk::", "")[:48]
win = [(name(k), s, e, is_dom(k)) for k, s, e in sel]
dd = sorted((s, e) for _, s, e, d in win if d)
# union of PATTERN intervals
merged = []
for s, e in dd:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
busy = sum(e - s for s, e in merged)
print(f"proof {which}: window {(w1 - w0) / 1e6:.2f} ms, '{pat}' running {busy / 1e6:.2f} ms = {busy / (w1 - w0):.3f}, {len(dd)} dispatches, {len(win)} kernels")
edges = [(w0, merged[0][0])] + [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)] + [(merged[-1][1], w1)]
tot_gap = 0
for i, (a, b) in enumerate(edges):
    if i:
        s, e = merged[i - 1]
        n = sum(1 for x, _ in dd if s <= x < e)
        print(f"  {(s - w0) / 1e6:8.2f} .. {(e - w0) / 1e6:8.2f}  BUSY {(e - s) / 1e6:7.2f} ms  ({n} launches)")
    if b - a < 20e3:
        continue
    tot_gap += b - a
    inside = {}
    idle, cur = 0, a
    for k, s, e, d in sorted(win, key=lambda r: r[1]):
        if d or e <= a or s >= b:
            continue
        ov = min(e, b) - max(s, a)
        v = inside.setdefault(k, [0, 0])
        v[0] += ov
        v[1] += 1
        if max(s, a) > cur:
            idle += max(s, a) - cur
        cur = max(cur, min(e, b))
    idle += max(0, b - cur)
    items = sorted(inside.items(), key=lambda kv: -kv[1][0])[:7]
    print(f"  {(a - w0) / 1e6:8.2f} .. {(b - w0) / 1e6:8.2f}  gap  {(b - a) / 1e6:7.2f} ms  idle {idle / 1e6:5.2f} | " +
          ", ".join(f"{k} x{c} {t / 1e6:.2f}" for k, (t, c) in items))
print(f"gaps: {tot_gap / 1e6:.2f} ms")
if os.environ.get("TIMELINE_DUMP"):   # every kernel of the proof in start order: start (ms after the window's start), duration (ms), name
    with open(os.environ["TIMELINE_DUMP"], "w") as f:
        for k, s, e, d in sorted(win, key=lambda r: r[1]):
            f.write(f"{(s - w0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {'*' if d else ' '} {k}\n")

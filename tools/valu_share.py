#!/usr/bin/env python3
"""Where a proof's VALU instructions go: per-kernel SQ_INSTS_VALU totals of one rocprofv3 --pmc pass, per proof.

    python tools/valu_share.py DB [PROOFS]

DB is the results.db of `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... -- python bench.py --steps 1 --warmup 0 ...`
(tools/profile_round.sh).  Kernels that only run at registration / set-up (window tables, twiddles, key generation) are listed
apart.  PROOFS defaults to the number of k_accumulate_u2 dispatches (one G2 MSM per proof)."""
import sqlite3
import sys

SETUP = ("k_fixed_base", "k_dbl_c", "k_sw_to_te_niels", "k_twiddle_table", "k_convert_to_u", "k_batch_to_affine", "k_table_")
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select kernel_name, count(*), sum(value), avg(duration) from counters_collection "
                       "where counter_name = 'SQ_INSTS_VALU' group by kernel_name order by 3 desc"))
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else next((n for k, n, _, _ in rows if k.startswith("czk::k_accumulate_u2(")), 1)
step = [(k, n, s, d) for k, n, s, d in rows if not any(t in k for t in SETUP)]
setup = [(k, n, s, d) for k, n, s, d in rows if any(t in k for t in SETUP)]
total = sum(s for _, _, s, _ in step)
print(f"# SQ_INSTS_VALU per proof ({proofs} proofs in the pass); wave-level instruction counts, share of the per-proof total {total / proofs:.4g}")
print(f"{'kernel':66s} {'calls/proof':>11s} {'insts/proof':>12s} {'share':>6s} {'avg_ms':>8s}")
for k, n, s, d in step:
    if s / total < 0.0005:
        continue
    print(f"{k[:66]:66s} {n / proofs:11.1f} {s / proofs:12.4g} {100 * s / total:5.1f}% {d / 1e6:8.3f}")
print("# set-up kernels (not part of a proof):")
for k, n, s, d in setup:
    print(f"{k[:66]:66s} {n:11d} {s:12.4g}")

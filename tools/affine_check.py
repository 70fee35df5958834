#!/usr/bin/env python3
"""Independent big-integer check of tools/affine_bench.bin's dump: every sampled slot must hold P1 + P2 by the affine
chord formula, in the unsaturated residue system (coordinate * 2^392 mod p, value < 3 p)."""
import sys

import numpy as np

Q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
RP_INV = pow(1 << 392, -1, Q)


def main(path, samples=4000):
    raw = np.fromfile(path, dtype=np.uint32)
    K, T = int(raw[0]), int(raw[2])
    body = raw[8:].reshape(3, K, 6, T, 4)          # (h1 | h2 | out), slot, chunk, thread, word

    def coord(a, j, t, c0):
        words = np.concatenate([body[a, j, c0 + c, t] for c in range(3)])
        return sum(int(w) << (32 * i) for i, w in enumerate(words))
    rng = np.random.default_rng(1)
    bad = 0
    for _ in range(samples):
        j, t = int(rng.integers(K)), int(rng.integers(T))
        X1, Y1, X2, Y2, X3, Y3 = (coord(a, j, t, c) for a in (0, 1, 2) for c in (0, 3))
        x1, y1, x2, y2 = (v * RP_INV % Q for v in (X1, Y1, X2, Y2))
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
        x3 = (lam * lam - x1 - x2) % Q
        y3 = (lam * (x1 - x3) - y1) % Q
        ok = X3 * RP_INV % Q == x3 and Y3 * RP_INV % Q == y3 and X3 < 3 * Q and Y3 < 3 * Q
        bad += 0 if ok else 1
    print(f"affine_check: K={K} T={T} sampled {samples} slots, {bad} wrong")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))

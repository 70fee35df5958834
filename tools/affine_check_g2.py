#!/usr/bin/env python3
"""Independent big-integer check of tools/affine_bench_g2.bin's dump: every sampled chain must hold acc + table[idx] by the affine
chord formula over Fq2 = Fq[u] / (u^2 + 5), coordinates in the unsaturated residue system (value * 2^392 mod p as 14 limbs of 28 bits)."""
import sys

import numpy as np

Q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
RP_INV = pow(1 << 392, -1, Q)


def mix(x):
    x &= 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def f2mul(a, b):
    return ((a[0] * b[0] - 5 * a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2inv(a):
    n = pow((a[0] * a[0] + 5 * a[1] * a[1]) % Q, -1, Q)
    return (a[0] * n % Q, -a[1] * n % Q)


def f2sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def main(path, samples=3000):
    raw = np.fromfile(path, dtype=np.uint32)
    K, T, mask = int(raw[0]), int(raw[2]), int(raw[4])
    n_acc = K * 16 * T * 4
    acc0 = raw[8:8 + n_acc].reshape(K, 16, T, 4)
    acc1 = raw[8 + n_acc:8 + 2 * n_acc].reshape(K, 16, T, 4)
    tab = raw[8 + 2 * n_acc:].reshape(mask + 1, 16, 4)

    def limbs(words):
        return sum(int(w) << (28 * i) for i, w in enumerate(words[:14]))

    def acc_fq(a, j, t, c0):
        return limbs(np.concatenate([a[j, c0 + c, t] for c in range(4)]))

    def tab_fq(e, c0):
        return limbs(tab[e, c0:c0 + 4].reshape(-1))
    rng = np.random.default_rng(1)
    bad = 0
    for _ in range(samples):
        j, t = int(rng.integers(K)), int(rng.integers(T))
        e = mix(t * 64 + j) & mask
        X1, Y1 = [tuple(acc_fq(acc0, j, t, c) * RP_INV % Q for c in (o, o + 4)) for o in (0, 8)]
        X2, Y2 = [tuple(tab_fq(e, c) * RP_INV % Q for c in (o, o + 4)) for o in (0, 8)]
        G = [tuple(acc_fq(acc1, j, t, c) for c in (o, o + 4)) for o in (0, 8)]
        lam = f2mul(f2sub(Y2, Y1), f2inv(f2sub(X2, X1)))
        x3 = f2sub(f2sub(f2mul(lam, lam), X1), X2)
        y3 = f2sub(f2mul(lam, f2sub(X1, x3)), Y1)
        ok = all(G[0][i] * RP_INV % Q == x3[i] and G[1][i] * RP_INV % Q == y3[i] and G[0][i] < 3 * Q and G[1][i] < 3 * Q for i in (0, 1))
        bad += 0 if ok else 1
    print(f"affine_check_g2: K={K} T={T} sampled {samples} chains, {bad} wrong")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))

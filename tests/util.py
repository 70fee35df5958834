"""Shared helpers for the parity tests: seeded inputs (SplitMix64) and limb plumbing."""
import numpy as np

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
_MASK = (1 << 64) - 1


def splitmix_u64(seed: int, n: int) -> np.ndarray:
    """n SplitMix64 outputs as uint64 (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_R_LIMBS = np.array([(R_MOD >> (64 * i)) & _MASK for i in range(4)], dtype=np.uint64)


def _lt_modulus(a: np.ndarray) -> np.ndarray:
    lt = np.zeros(a.shape[0], dtype=bool)
    eq = np.ones(a.shape[0], dtype=bool)
    for j in (3, 2, 1, 0):
        lt |= eq & (a[:, j] < _R_LIMBS[j])
        eq &= a[:, j] == _R_LIMBS[j]
    return lt


def rand_fr_canonical(seed: int, n: int) -> np.ndarray:
    """(n,4) uint64 canonical values < r: 4 limbs per candidate, top 3 bits masked (REPR_SHAVE_BITS, fr.rs:44),
    rejection of candidates >= r (fields/arithmetic.rs:199-214)."""
    out = np.zeros((0, 4), dtype=np.uint64)
    chunk = 0
    while out.shape[0] < n:
        m = max(16, int((n - out.shape[0]) * 1.7) + 8)
        raw = splitmix_u64(seed + 0x1000003 * chunk, 4 * m).reshape(m, 4)
        raw[:, 3] &= np.uint64(_MASK >> 3)
        out = np.vstack([out, raw[_lt_modulus(raw)]])
        chunk += 1
    return np.ascontiguousarray(out[:n])


def ints_to_limbs(vals, n_limbs):
    out = np.zeros((len(vals), n_limbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(n_limbs):
            out[i, j] = (v >> (64 * j)) & _MASK
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    arr = arr.reshape(-1, arr.shape[-1])
    return [sum(int(arr[i, j]) << (64 * j) for j in range(arr.shape[1])) for i in range(arr.shape[0])]


def dot_mod_r(k, s) -> int:
    """sum_i k_i * s_i mod r for (n,4) uint64 limb arrays, exactly and without per-element Python integers: 16-bit limbs as
    float64, the 16 x 16 limb-pair sums by dgemm in chunks of 2^18 rows (< 2^50: exact), accumulated as Python ints."""
    tot = [[0] * 16 for _ in range(16)]
    n = k.shape[0]
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        K = np.ascontiguousarray(k[lo:hi]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        S = np.ascontiguousarray(s[lo:hi]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        m = K.T @ S
        for a in range(16):
            for b in range(16):
                tot[a][b] += int(m[a, b])
    return sum(tot[a][b] << (16 * (a + b)) for a in range(16) for b in range(16)) % R_MOD

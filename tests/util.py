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


def child_errors(stderr: str, limit: int = 3000) -> str:
    """What a failed multi-rank child actually raised: the traceback blocks of its stderr (torchrun's summary at the end of stderr names the rank but
    not the exception), else the tail."""
    lines = stderr.splitlines()
    keep = []
    for i, ln in enumerate(lines):
        if "Traceback (most recent call last)" in ln:
            keep += lines[i:i + 30] + ["..."]
    text = "\n".join(keep) if keep else stderr
    return text[-limit:]


def run_ranks(cmd, retries: int = 1, **kw):
    """subprocess.run for a command that starts several ranks (torchrun / fork): a failed attempt is reported on stderr with the exception its rank
    raised, and the command is tried once more -- rendezvous on the GPU boxes fails now and then for reasons outside the repository (a resolver that
    times out: "[c10d] The hostname of the client socket cannot be retrieved"), and the driver's `pytest -x` must not stop on that.  A defect fails
    twice."""
    import subprocess
    import sys
    kw.setdefault("capture_output", True)
    kw.setdefault("text", True)
    for attempt in range(retries + 1):
        r = subprocess.run(cmd, **kw)
        if r.returncode == 0 or attempt == retries:
            return r
        print(f"[run_ranks] attempt {attempt + 1} of {' '.join(map(str, cmd[:6]))} ... failed (rc {r.returncode}); its ranks raised:\n{child_errors(r.stderr)}\nretrying",
              file=sys.stderr, flush=True)
    return r

"""Full-size answers of the CPU checker as committed digests.

The GPU suite compares the HIP path with the checker (oracle/) at BASELINE sizes: 2^17 .. 2^21-point NTTs on four lanes, MSMs of
2^20 + 1 points against the checker's own Pippenger, the complete 2^20-constraint Groth16 step.  The checker needs minutes of host
time for those answers, which made the suite's run time depend on the box's host cores.  The answers are deterministic functions of
seeded inputs, so they are computed ONCE by `tests/golden/make_fullsize.py` -- from oracle/ alone, no GPU, bases included (the
checker's restatement of the reference's FixedBaseMSM) -- and committed as SHA-256 digests in tests/golden/fullsize_digests.json.
A test then hashes what the GPU produced and compares; every comparison the suite made before is still made, byte for byte
(equal digests <=> equal bytes).  To catch drift on either side, ONE case per pytest session, chosen at random (or by
CZK_FULLSIZE_LIVE=<case id>|all|none), is ALSO recomputed live by the checker and compared with the GPU's bytes directly, and the
CPU suite regenerates a subset of the file and compares it with the committed one (tests/test_golden.py).

A case is a function `expected(orc) -> {part name: bytes}`; its digest entry is {part name: sha256 hex}."""
import hashlib
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DIGEST_FILE = os.path.join(HERE, "golden", "fullsize_digests.json")
if HERE not in sys.path:
    sys.path.insert(0, HERE)
from util import ints_to_limbs, rand_fr_canonical  # noqa: E402

KINDS = ("fft", "ifft", "coset_fft", "coset_ifft")     # = CZK_FFT .. CZK_COSET_IFFT = orc.FFT .. orc.COSET_IFFT
BASE_SEED = 0xBA5E5


# ---- inputs (the tests build the same ones) ----------------------------------------------------------------------------------
def ntt_input(orc, log_d, lanes=4):
    d = 1 << log_d
    return orc.fr_from_repr(rand_fr_canonical(4242 + log_d, lanes * d)).reshape(lanes, d, 4)


def ntt_in_len(kind: int, d: int) -> int:
    """forward kinds on the full domain, inverse kinds on a ragged prefix (resize(size, zero), radix2/mod.rs:100-101)"""
    return d if kind in (0, 2) else d - 12345


_bases_cache = {}


def oracle_bases(orc, g, seed_offset, n, first_inf=False):
    """[k_i] G for k = rand_fr_canonical(BASE_SEED + seed_offset, n) by the checker's FixedBaseMSM (the reference generator's algorithm,
    algebra/ec/src/msm/fixed_base.rs): what ctx.fixed_base_points produces on the GPU.  Returns (bases, inf)."""
    key = (g, seed_offset, n)
    if key not in _bases_cache:
        pts, inf = orc.fixed_base_points(g, rand_fr_canonical(BASE_SEED + seed_offset, n))
        assert not inf.any()
        _bases_cache.clear()                           # one array at a time: 2^21 G1 points are 200 MB
        _bases_cache[key] = pts
    inf = np.zeros(n, dtype=np.uint8)
    if first_inf:
        inf[0] = 1
    return _bases_cache[key], inf


def msm_scalars(orc, n, lanes=4):
    """Montgomery scalars of the full-size Pippenger test: a unit scalar and a zero on lane 2 (variable_base.rs:19, 44-48)"""
    s = orc.fr_from_repr(rand_fr_canonical(0xFACE, lanes * n)).reshape(lanes, n, 4)
    s[2, 5] = orc.fr_from_repr(ints_to_limbs([1], 4))[0]
    s[2, 6] = 0
    return s


def affine_bytes(aff, inf) -> bytes:
    return np.ascontiguousarray(aff, np.uint64).tobytes() + bytes([int(bool(inf))])


# ---- cases ----------------------------------------------------------------------------------------------------------------------
def _ntt_case(log_d, kind):
    def expected(orc):
        x = ntt_input(orc, log_d)
        in_len = ntt_in_len(kind, 1 << log_d)
        return {f"lane{ln}": orc.ntt_fr(x[ln, :in_len], log_d, kind, in_len).tobytes() for ln in range(x.shape[0])}
    return expected


def _bases_case(g, seed_offset, n):
    def expected(orc):
        return {"points": oracle_bases(orc, g, seed_offset, n)[0].tobytes()}
    return expected


def _msm_case(g, n, lane):
    def expected(orc):
        bases, inf = oracle_bases(orc, g, 3, n, first_inf=True)
        s = msm_scalars(orc, n)
        aff, is_inf = orc.jac_to_affine(g, orc.multi_scalar_mul(g, bases, inf, s[lane]))
        return {"affine": affine_bytes(aff, is_inf)}
    return expected


GROTH16_QUERIES = (("h", 1, 1, False), ("l", 1, 2, False), ("a", 1, 3, False), ("b_g1", 1, 4, True), ("b_g2", 2, 5, True))


def groth16_query_len(name, N, D):
    return {"h": D - 1, "l": N, "a": N + 1, "b_g1": N + 1, "b_g2": N + 1}[name]


def _groth16_case(log_n):
    def expected(orc):
        from test_device_handles import beaver_shortcut_lanes, groth16_inputs
        N = 1 << log_n
        a0, b0, c0, wit, asg, log_d, _ = groth16_inputs(orc, N)
        D, L = 1 << log_d, a0.shape[0]
        keys, infs = {}, {}
        for name, g, sd, first_inf in GROTH16_QUERIES:
            b, inf = oracle_bases(orc, g, sd, groth16_query_len(name, N, D), first_inf)
            keys[name], infs[name] = b.copy(), inf
        a1, b1, c1 = beaver_shortcut_lanes(orc, a0, b0, c0)
        del a0, b0
        want = orc.groth16_local_par(log_d, N, a1, b1, c1, wit, asg, keys["h"], keys["l"], keys["a"], keys["b_g1"], keys["b_g2"], infs["b_g1"],
                                     threads=min(orc.max_threads(), 32))
        out = {f"hvec_lane{ln}": a1[ln].tobytes() for ln in range(L)}       # a1 ends as h (in place): the quotient's coefficients on every lane
        for q, (name, g, _, _) in enumerate(GROTH16_QUERIES):
            w = 18 if g == 1 else 36
            for ln in range(L):
                aff, is_inf = orc.jac_to_affine(g, want[ln, 18 * q:18 * q + w])
                out[f"{name}_lane{ln}"] = affine_bytes(aff, is_inf)
        return out
    return expected


FULL_N = (1 << 20) + 1
CASES = {}
for _log_d in (17, 18, 19, 20, 21):
    for _k, _name in enumerate(KINDS):
        CASES[f"ntt_2e{_log_d}_{_name}"] = _ntt_case(_log_d, _k)
CASES["bases_g1_seed3_2e20p1"] = _bases_case(1, 3, FULL_N)
CASES["bases_g2_seed3_2e20p1"] = _bases_case(2, 3, FULL_N)
CASES["msm_g1_2e20p1_lane0"] = _msm_case(1, FULL_N, 0)
CASES["msm_g1_2e20p1_lane3"] = _msm_case(1, FULL_N, 3)
CASES["msm_g2_2e20p1_lane1"] = _msm_case(2, FULL_N, 1)
for _name, _g, _sd, _ in GROTH16_QUERIES:
    CASES[f"bases_groth16_{_name}_2e20"] = _bases_case(_g, _sd, groth16_query_len(_name, 1 << 20, 1 << 21))
CASES["groth16_spdz2_2e20"] = _groth16_case(20)
# what the CPU suite regenerates on every run (tests/test_golden.py): seconds each on 8 cores
CHEAP = ("ntt_2e17_fft", "ntt_2e17_ifft", "ntt_2e17_coset_fft", "ntt_2e17_coset_ifft", "ntt_2e18_coset_ifft", "bases_g1_seed3_2e20p1")


def digest_parts(parts: dict) -> dict:
    return {k: hashlib.sha256(v).hexdigest() for k, v in parts.items()}


def load() -> dict:
    with open(DIGEST_FILE) as f:
        return json.load(f)["cases"]


_live = None
_live_results = {}     # the live case's recomputed parts (two hosts may be checked against the same case)
live_log = []          # (case id, "live") of what this session recomputed: tests may assert / report on it


def live_case() -> str:
    """The case this pytest session recomputes with the checker: CZK_FULLSIZE_LIVE = a case id, `all`, `none`; default: one at random"""
    global _live
    if _live is None:
        _live = os.environ.get("CZK_FULLSIZE_LIVE") or random.SystemRandom().choice(sorted(CASES))
    return _live


def expect(case_id: str, got: dict, orc, parts=None):
    """Asserts that the GPU's bytes `got` ({part: bytes}) are the checker's: by the committed digests always, and by a live run of the
    checker when this case is the session's live one.  `parts`: the subset of the case's parts this caller produces (default all)."""
    want = load()[case_id]
    names = list(parts) if parts is not None else list(want)
    assert set(names) <= set(want) and set(names) <= set(got), (case_id, sorted(want), sorted(got))
    for k in names:
        assert hashlib.sha256(got[k]).hexdigest() == want[k], f"{case_id}/{k}: the GPU result differs from the checker's committed digest"
    if live_case() in (case_id, "all"):
        if case_id not in _live_results:
            _live_results.clear()
            _live_results[case_id] = CASES[case_id](orc)
        live = _live_results[case_id]
        for k in names:
            assert got[k] == live[k], f"{case_id}/{k}: the GPU result differs from the checker's live result"
        assert digest_parts(live) == want, f"{case_id}: the checker's live result differs from its committed digest (regenerate tests/golden/fullsize_digests.json?)"
        live_log.append(case_id)
        print(f"[fullsize] {case_id}: recomputed live by the checker, equal")


def generate(only=None, verbose=True) -> dict:
    import time
    import orc
    out = {}
    for cid in (only or CASES):
        t0 = time.time()
        out[cid] = digest_parts(CASES[cid](orc))
        if verbose:
            print(f"{cid}: {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    return out

"""ctypes binding of oracle/libczk_oracle.so (TEST INFRASTRUCTURE ONLY -- the CPU checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays are numpy uint64, little-endian limbs, Montgomery form unless a name says `repr`/`canonical`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libczk_oracle.so")

FFT, IFFT, COSET_FFT, COSET_IFFT = 0, 1, 2, 3


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("czk_oracle.c", "fp_tmpl.h", "ec_tmpl.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


# ----------------------------------------------------------------------------- int <-> limbs
def ints_to_limbs(vals, n_limbs):
    out = np.zeros((len(vals), n_limbs), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        for j in range(n_limbs):
            out[i, j] = (v >> (64 * j)) & mask
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    arr = arr.reshape(-1, arr.shape[-1])
    return [sum(int(arr[i, j]) << (64 * j) for j in range(arr.shape[1])) for i in range(arr.shape[0])]


# ----------------------------------------------------------------------------- field ops
def _binop(name, width):
    def f(a, b):
        a, b = _u64(a), _u64(b)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(b), _p(out), C.c_size_t(a.size // width))
        return out
    return f


def _unop(name, width):
    def f(a):
        a = _u64(a)
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(out), C.c_size_t(a.size // width))
        return out
    return f


fr_mul, fr_add, fr_sub = _binop("orc_fr_mul", 4), _binop("orc_fr_add", 4), _binop("orc_fr_sub", 4)
fr_sqr, fr_neg, fr_dbl, fr_inv = (_unop("orc_fr_sqr", 4), _unop("orc_fr_neg", 4), _unop("orc_fr_dbl", 4),
                                  _unop("orc_fr_inv", 4))
fq_mul, fq_add, fq_sub = _binop("orc_fq_mul", 6), _binop("orc_fq_add", 6), _binop("orc_fq_sub", 6)
fq_sqr, fq_neg, fq_dbl, fq_inv = (_unop("orc_fq_sqr", 6), _unop("orc_fq_neg", 6), _unop("orc_fq_dbl", 6),
                                  _unop("orc_fq_inv", 6))
fq2_mul, fq2_add, fq2_sub = _binop("orc_fq2_mul", 12), _binop("orc_fq2_add", 12), _binop("orc_fq2_sub", 12)
fq2_sqr, fq2_inv = _unop("orc_fq2_sqr", 12), _unop("orc_fq2_inv", 12)
fr_into_repr = _unop("orc_fr_into_repr", 4)
fq_into_repr = _unop("orc_fq_into_repr", 6)


def fr_from_repr(a):
    a = _u64(a)
    out = np.empty_like(a)
    rc = lib().orc_fr_from_repr(_p(a), _p(out), C.c_size_t(a.size // 4))
    if rc:
        raise ValueError("from_repr: value >= modulus")
    return out


def fq_from_repr(a):
    a = _u64(a)
    out = np.empty_like(a)
    rc = lib().orc_fq_from_repr(_p(a), _p(out), C.c_size_t(a.size // 6))
    if rc:
        raise ValueError("from_repr: value >= modulus")
    return out


# ----------------------------------------------------------------------------- NTT
def ntt_fr(data, log_d, kind, in_len=None):
    """In-order {fft, ifft, coset_fft, coset_ifft} of one Fr lane; returns a new (D, 4) array."""
    d = 1 << log_d
    data = _u64(data).reshape(-1, 4)
    if in_len is None:
        in_len = data.shape[0]
    buf = np.zeros((d, 4), dtype=np.uint64)
    buf[:in_len] = data[:in_len]
    rc = lib().orc_ntt_fr(_p(buf), C.c_uint(log_d), C.c_int(kind), C.c_size_t(in_len))
    if rc:
        raise ValueError("orc_ntt_fr failed")
    return buf


def domain_constants(log_d):
    out = np.zeros((6, 4), dtype=np.uint64)
    rc = lib().orc_domain_constants(C.c_uint(log_d), _p(out))
    if rc:
        raise ValueError("domain too large")
    return dict(zip(["size_inv", "group_gen", "group_gen_inv", "generator", "generator_inv", "vanishing_inv"], out))


def fr_horner(coeffs, x):
    coeffs = _u64(coeffs).reshape(-1, 4)
    x = _u64(x)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_horner(_p(coeffs), C.c_size_t(coeffs.shape[0]), _p(x), _p(out))
    return out


# ----------------------------------------------------------------------------- groups
def _grp(g):
    aff_w = 12 if g == 1 else 24
    jac_w = 18 if g == 1 else 36
    return aff_w, jac_w


def msm(g, bases, inf, scalars_canonical):
    """VariableBaseMSM::multi_scalar_mul: canonical scalars (n,4); returns Jacobian limbs."""
    aff_w, jac_w = _grp(g)
    bases = _u64(bases).reshape(-1, aff_w)
    scalars = _u64(scalars_canonical).reshape(-1, 4)
    n = min(bases.shape[0], scalars.shape[0])
    inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(jac_w, dtype=np.uint64)
    getattr(lib(), f"orc_g{g}_msm")(_p(bases), _p(inf), _p(scalars), C.c_size_t(n), _p(out))
    return out


def multi_scalar_mul(g, bases, inf, scalars_mont):
    """AffineCurve::multi_scalar_mul: Montgomery scalars; lengths may differ (min is used)."""
    aff_w, jac_w = _grp(g)
    bases = _u64(bases).reshape(-1, aff_w)
    scalars = _u64(scalars_mont).reshape(-1, 4)
    inf = np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(jac_w, dtype=np.uint64)
    getattr(lib(), f"orc_g{g}_multi_scalar_mul")(_p(bases), _p(inf), _p(scalars), C.c_size_t(bases.shape[0]),
                                                 C.c_size_t(scalars.shape[0]), _p(out))
    return out


def jac_to_affine(g, jac):
    aff_w, _ = _grp(g)
    jac = _u64(jac)
    out = np.zeros(aff_w, dtype=np.uint64)
    fn = getattr(lib(), f"orc_g{g}_jac_to_affine")
    fn.restype = C.c_int
    is_inf = fn(_p(jac), _p(out))
    return out, bool(is_inf)


def scalar_mul(g, base_aff, base_inf, k_canonical):
    _, jac_w = _grp(g)
    out = np.zeros(jac_w, dtype=np.uint64)
    base_aff, k = _u64(base_aff), _u64(k_canonical)
    getattr(lib(), f"orc_g{g}_scalar_mul")(_p(base_aff), C.c_int(int(base_inf)), _p(k), _p(out))
    return out


def jac_add(g, a, b):
    _, jac_w = _grp(g)
    out = np.zeros(jac_w, dtype=np.uint64)
    a, b = _u64(a), _u64(b)
    getattr(lib(), f"orc_g{g}_jac_add")(_p(a), _p(b), _p(out))
    return out


def jac_add_mixed(g, a, b_aff, b_inf=False):
    _, jac_w = _grp(g)
    out = np.zeros(jac_w, dtype=np.uint64)
    a, b_aff = _u64(a), _u64(b_aff)
    getattr(lib(), f"orc_g{g}_jac_add_mixed")(_p(a), _p(b_aff), C.c_int(int(b_inf)), _p(out))
    return out


def jac_double(g, a):
    _, jac_w = _grp(g)
    out = np.zeros(jac_w, dtype=np.uint64)
    a = _u64(a)
    getattr(lib(), f"orc_g{g}_jac_double")(_p(a), _p(out))
    return out


def g1_on_curve(aff):
    aff = _u64(aff)
    fn = lib().orc_g1_on_curve
    fn.restype = C.c_int
    return bool(fn(_p(aff)))


def g2_on_curve(aff, coeff_b):
    aff, coeff_b = _u64(aff), _u64(coeff_b)
    fn = lib().orc_g2_on_curve
    fn.restype = C.c_int
    return bool(fn(_p(aff), _p(coeff_b)))


# ----------------------------------------------------------------------------- witness map
def witness_map_plain(a, b, c, log_d):
    """Single-prover R1CStoQAP::witness_map on one Fr lane; returns h (D,4)."""
    a, b, c = (np.array(_u64(x).reshape(-1, 4)) for x in (a, b, c))
    lib().orc_witness_map_plain(_p(a), _p(b), _p(c), C.c_uint(log_d))
    return a


def witness_map_pre(a, b, log_d):
    a, b = (np.array(_u64(x).reshape(-1, 4)) for x in (a, b))
    lib().orc_witness_map_pre(_p(a), _p(b), C.c_uint(log_d))
    return a, b


def witness_map_post(ab, c, log_d):
    ab, c = (np.array(_u64(x).reshape(-1, 4)) for x in (ab, c))
    lib().orc_witness_map_post(_p(ab), _p(c), C.c_uint(log_d))
    return ab


# ----------------------------------------------------------------------------- callers either side of the NTT
def r1cs_matvec(row_ptr, col, coeff, z):
    """evaluate_constraint over every row: row_ptr (m+1,) u64, col (nnz,) u32, coeff (nnz,4), z (n_vars,4) -> (m,4)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    coeff = np.ascontiguousarray(_u64(coeff).reshape(-1, 4))
    z = np.ascontiguousarray(_u64(z).reshape(-1, 4))
    m = len(row_ptr) - 1
    out = np.zeros((m, 4), dtype=np.uint64)
    lib().orc_r1cs_matvec(_p(row_ptr), col.ctypes.data_as(C.c_void_p), _p(coeff), C.c_size_t(m), _p(z), _p(out))
    return out


def poly_div_linear(coeffs, z):
    """coeffs (n,4) / (X - z) -> (quotient (n-1,4), remainder (4,))."""
    coeffs = np.ascontiguousarray(_u64(coeffs).reshape(-1, 4))
    z = np.ascontiguousarray(_u64(z).reshape(4))
    n = len(coeffs)
    q = np.zeros((max(n - 1, 0), 4), dtype=np.uint64)
    r = np.zeros(4, dtype=np.uint64)
    qbuf = q if len(q) else np.zeros((1, 4), dtype=np.uint64)
    lib().orc_poly_div_linear(_p(coeffs) if n else None, C.c_size_t(n), _p(z), _p(qbuf), _p(r))
    return q, r


def fr_prefix_product(x):
    x = np.ascontiguousarray(_u64(x).reshape(-1, 4))
    out = np.zeros_like(x)
    if len(x):
        lib().orc_fr_prefix_product(_p(x), C.c_size_t(len(x)), _p(out))
    return out


def fr_batch_inverse(v, coeff):
    v = np.array(_u64(v).reshape(-1, 4))
    coeff = np.ascontiguousarray(_u64(coeff).reshape(4))
    if len(v):
        lib().orc_fr_batch_inverse(_p(v), C.c_size_t(len(v)), _p(coeff))
    return v


# ----------------------------------------------------------------------------- all-host-cores CPU baseline (bench.py)
def max_threads() -> int:
    fn = lib().orc_max_threads
    fn.restype = C.c_int
    return int(fn())


def generator_affine(g):
    """Montgomery limbs of the G1 / G2 generator (curves/bls12_377/src/curves/g1.rs:46-51, g2.rs:64-86)."""
    from pyref import G1_GEN, G2_GEN, fq_to_mont
    if g == 1:
        return ints_to_limbs([fq_to_mont(G1_GEN[0]), fq_to_mont(G1_GEN[1])], 6).reshape(-1)
    return ints_to_limbs([fq_to_mont(G2_GEN[0][0]), fq_to_mont(G2_GEN[0][1]), fq_to_mont(G2_GEN[1][0]), fq_to_mont(G2_GEN[1][1])], 6).reshape(-1)


def chain_points(g, n):
    """n distinct subgroup points P_0 = G, P_{i+1} = 2 P_i + G (affine): cheap bases for timing runs."""
    aff_w, _ = _grp(g)
    gen = _u64(generator_affine(g))
    out = np.zeros((n, aff_w), dtype=np.uint64)
    getattr(lib(), f"orc_g{g}_chain_points")(_p(gen), _p(out), C.c_size_t(n))
    return out


def fixed_base_points(g, k_canonical, threads=0):
    """FixedBaseMSM::multi_scalar_mul over the group's generator (algebra/ec/src/msm/fixed_base.rs:11-96): (n, 12|24) affine Montgomery
    limbs and (n,) infinity bytes of [k_i] G -- the reference generator's way of building a key's queries; all host threads."""
    aff_w, _ = _grp(g)
    k = _u64(k_canonical).reshape(-1, 4)
    n = k.shape[0]
    gen = _u64(generator_affine(g))
    out = np.zeros((n, aff_w), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    getattr(lib(), f"orc_g{g}_fixed_base_msm")(_p(gen), _p(k), C.c_size_t(n), _p(out), _p(inf), C.c_int(threads))
    return out, inf


def groth16_local_par(log_d, N, a, b, c, wit, asg, h_q, l_q, a_q, b1_q, b2_q, inf_b, threads=0):
    """One proof's local compute (witness map + 5 MSMs per share lane) with OpenMP tasks on `threads` host threads
    (0 = all).  a, b, c: (lanes, D, 4) and are overwritten; returns (lanes, 108) u64: h, l, a, b_g1 Jacobian (18 each), b_g2 (36)."""
    a, b, c = _u64(a), _u64(b), _u64(c)
    lanes = a.shape[0]
    wit, asg = _u64(wit), _u64(asg)
    h_q, l_q, a_q, b1_q, b2_q = (_u64(x) for x in (h_q, l_q, a_q, b1_q, b2_q))
    inf0 = np.zeros(max(h_q.shape[0], N + 1), dtype=np.uint8)
    inf_b = np.ascontiguousarray(inf_b, dtype=np.uint8)
    out = np.zeros((lanes, 108), dtype=np.uint64)
    lib().orc_groth16_local_par(C.c_uint(log_d), C.c_size_t(N), C.c_size_t(lanes), _p(a), _p(b), _p(c), _p(wit), _p(asg), _p(h_q), _p(l_q),
                                _p(a_q), _p(b1_q), _p(b2_q), _p(inf0), _p(inf_b), _p(out), C.c_int(threads))
    return out


# ----------------------------------------------------------------------------- GSZ / Shamir shares (gsz20/mod.rs)
def fr_root_of_unity_mixed(n):
    """F::get_root_of_unity(n) for n = 2^a * 3^b (fields/mod.rs:337-367); None when no such subgroup exists."""
    out = np.zeros(4, dtype=np.uint64)
    fn = lib().orc_fr_root_of_unity_mixed
    fn.restype = C.c_int
    return None if fn(C.c_size_t(n), _p(out)) else out


def gsz_open(shares, degree=0, degrees=None):
    """batch_open of GSZ shares: shares (parties, n, 4) -> (values (n,4), number of degree-bound violations)."""
    shares = _u64(shares)
    parties, n = shares.shape[0], shares.shape[1]
    out = np.zeros((n, 4), dtype=np.uint64)
    bad = C.c_uint64(0)
    dg = None if degrees is None else np.ascontiguousarray(degrees, dtype=np.uint32)
    fn = lib().orc_gsz_open
    fn.restype = C.c_int
    rc = fn(_p(shares), C.c_size_t(parties), C.c_size_t(n), _p(dg) if dg is not None else C.c_void_p(0), C.c_uint(degree), _p(out), C.byref(bad))
    if rc:
        raise ValueError(f"no evaluation domain of size {parties}")
    return out, bad.value


def gsz_share(coeffs, parties):
    """(parties, 4): p(w^j) for the polynomial with Montgomery coefficients `coeffs` (k, 4)."""
    coeffs = _u64(coeffs).reshape(-1, 4)
    out = np.zeros((parties, 4), dtype=np.uint64)
    fn = lib().orc_gsz_share
    fn.restype = C.c_int
    if fn(_p(coeffs), C.c_size_t(coeffs.shape[0]), C.c_size_t(parties), _p(out)):
        raise ValueError(f"no evaluation domain of size {parties}")
    return out


def ntt_fr_mixed(data, size, kind, in_len=None):
    """MixedRadixEvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place over a domain of `size` = 2^a or 3 * 2^a."""
    data = _u64(data).reshape(-1, 4)
    if in_len is None:
        in_len = data.shape[0]
    buf = np.zeros((size, 4), dtype=np.uint64)
    buf[:in_len] = data[:in_len]
    fn = lib().orc_ntt_fr_mixed
    fn.restype = C.c_int
    if fn(_p(buf), C.c_size_t(size), C.c_int(kind), C.c_size_t(in_len)):
        raise ValueError("no mixed-radix domain of that size (or in_len > size)")
    return buf

"""Pins the oracles on what the reference itself holds for this path:
constant self-consistency KATs, the G1 generator KAT, and algebraic identities mirrored from the
reference's own tests.  CPU only."""
import random

import numpy as np
import pyref as P


def tonelli(n, p):
    assert pow(n, (p - 1) // 2, p) == 1
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(n, q, p), pow(n, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c, t, r = i, b * b % p, t * b * b % p, r * b % p
    return r


def test_field_constants_self_consistent():
    # fr.rs:33-63, fq.rs:26-62
    assert P.limbs_to_int(P.FR_MODULUS_LIMBS) == P.R_MOD and P.limbs_to_int(P.FQ_MODULUS_LIMBS) == P.Q_MOD
    assert (P.R_MOD * P.FR_INV) % 2**64 == 2**64 - 1 and (P.Q_MOD * P.FQ_INV) % 2**64 == 2**64 - 1
    assert P.limbs_to_int(P.FR_R_LIMBS) == 2**256 % P.R_MOD
    assert P.limbs_to_int(P.FQ_R_LIMBS) == 2**384 % P.Q_MOD
    assert P.limbs_to_int(P.FR_R2_LIMBS) == pow(2, 512, P.R_MOD)
    assert P.limbs_to_int(P.FQ_R2_LIMBS) == pow(2, 768, P.Q_MOD)
    assert P.R_MOD.bit_length() == 253 and P.Q_MOD.bit_length() == 377  # tests.rs:346-349
    assert P.R_MOD - 1 == P.FR_T << 47 and P.Q_MOD - 1 == P.FQ_T << 46


def test_generators_and_roots_of_unity():
    # fr.rs:69-74 decodes to 22 (comment says 11); fq.rs GENERATOR = -5
    assert P.fr_from_mont(P.limbs_to_int(P.FR_GENERATOR_LIMBS)) == 22
    assert P.fq_from_mont(P.limbs_to_int(P.FQ_GENERATOR_LIMBS)) == P.Q_MOD - 5
    # curves/bls12_377/src/fields/tests.rs:352-370: GENERATOR^T == TWO_ADIC_ROOT_OF_UNITY, root^(2^s) == 1
    two_adic_r = P.fr_from_mont(P.limbs_to_int(P.FR_TWO_ADIC_ROOT_LIMBS))
    assert pow(22, P.FR_T, P.R_MOD) == two_adic_r and pow(two_adic_r, 1 << 47, P.R_MOD) == 1
    two_adic_q = P.fq_from_mont(P.limbs_to_int(P.FQ_TWO_ADIC_ROOT_LIMBS))
    assert pow(P.Q_MOD - 5, P.FQ_T, P.Q_MOD) == two_adic_q and pow(two_adic_q, 1 << 46, P.Q_MOD) == 1
    # the root the reference actually uses (fields/mod.rs:360-367) is LARGE^3, NOT the two-adic root
    large = P.fr_from_mont(P.limbs_to_int(P.FR_LARGE_SUBGROUP_ROOT_LIMBS))
    assert pow(large, 3 << 47, P.R_MOD) == 1 and pow(large, 3, P.R_MOD) != two_adic_r
    w21 = P.fr_root_of_unity(21)
    assert pow(w21, 1 << 20, P.R_MOD) == P.R_MOD - 1
    assert P.fr_root_of_unity(48) is None  # radix2/mod.rs:61-63 -> None
    # tests.rs:382-395 legendre
    assert pow(4, (P.Q_MOD - 1) // 2, P.Q_MOD) == 1 and pow(5, (P.Q_MOD - 1) // 2, P.Q_MOD) == P.Q_MOD - 1


def test_g1_generator_kat():
    # curves/tests.rs:93-120: generator == cofactor * (x = 1, y = min(+-sqrt(x^3 + 1)))
    y = tonelli(2, P.Q_MOD)
    y = min(y, P.Q_MOD - y)
    assert P.ec_on_curve(P.F1, (1, y), 1)
    g = P.ec_mul(P.F1, P.G1_COFACTOR, (1, y))
    assert g == P.G1_GEN
    assert P.ec_mul(P.F1, P.R_MOD, P.G1_GEN) is P.INF
    assert P.ec_on_curve(P.F2, P.G2_GEN, P.G2_B) and P.ec_mul(P.F2, P.R_MOD, P.G2_GEN) is P.INF


def _rand_fr(rng, n):
    return [rng.randrange(P.R_MOD) for _ in range(n)]


def _rand_fq(rng, n):
    return [rng.randrange(P.Q_MOD) for _ in range(n)]


def test_c_oracle_fr_fq_limb_exact(orc):
    rng = random.Random(1)
    n = 300
    edge_r = [0, 1, P.R_MOD - 1, P.R_MOD - 2, 2, (1 << 252)]
    edge_q = [0, 1, P.Q_MOD - 1, P.Q_MOD - 2, 2, (1 << 376)]
    for mod, nl, mr, edge, pre in ((P.R_MOD, 4, P.FR_MONT_R, edge_r, "fr"), (P.Q_MOD, 6, P.FQ_MONT_R, edge_q, "fq")):
        a = edge + [rng.randrange(mod) for _ in range(n)]
        b = list(reversed(edge)) + [rng.randrange(mod) for _ in range(n)]
        A, B = orc.ints_to_limbs(a, nl), orc.ints_to_limbs(b, nl)   # treat as Montgomery residues directly
        rinv = pow(mr, -1, mod)
        assert orc.limbs_to_ints(getattr(orc, pre + "_mul")(A, B)) == [x * y * rinv % mod for x, y in zip(a, b)]
        assert orc.limbs_to_ints(getattr(orc, pre + "_sqr")(A)) == [x * x * rinv % mod for x in a]
        assert orc.limbs_to_ints(getattr(orc, pre + "_add")(A, B)) == [(x + y) % mod for x, y in zip(a, b)]
        assert orc.limbs_to_ints(getattr(orc, pre + "_sub")(A, B)) == [(x - y) % mod for x, y in zip(a, b)]
        assert orc.limbs_to_ints(getattr(orc, pre + "_neg")(A)) == [(-x) % mod for x in a]
        assert orc.limbs_to_ints(getattr(orc, pre + "_dbl")(A)) == [(2 * x) % mod for x in a]
        assert orc.limbs_to_ints(getattr(orc, pre + "_into_repr")(A)) == [x * rinv % mod for x in a]
        assert orc.limbs_to_ints(getattr(orc, pre + "_from_repr")(A)) == [x * mr % mod for x in a]
        nz = [x for x in a if x]
        NZ = orc.ints_to_limbs(nz, nl)
        # inverse of a Montgomery residue x = aR is a^-1 R = x^-1 R^2
        assert orc.limbs_to_ints(getattr(orc, pre + "_inv")(NZ)) == [pow(x, -1, mod) * mr * mr % mod for x in nz]


def test_c_oracle_fq2(orc):
    rng = random.Random(2)
    n = 100
    a = [(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD)) for _ in range(n)] + [(0, 5), (7, 0), (1, 0)]
    b = [(rng.randrange(P.Q_MOD), rng.randrange(P.Q_MOD)) for _ in range(n)] + [(3, 0), (0, 9), (0, 1)]
    tm = lambda v: [P.fq_to_mont(x) for t in v for x in t]
    A = orc.ints_to_limbs(tm(a), 6).reshape(-1, 12)
    B = orc.ints_to_limbs(tm(b), 6).reshape(-1, 12)
    un = lambda arr: [tuple(P.fq_from_mont(v) for v in orc.limbs_to_ints(row.reshape(2, 6))) for row in arr]
    assert un(orc.fq2_mul(A, B)) == [P.fq2_mul(x, y) for x, y in zip(a, b)]
    assert un(orc.fq2_sqr(A)) == [P.fq2_mul(x, x) for x in a]
    assert un(orc.fq2_add(A, B)) == [P.fq2_add(x, y) for x, y in zip(a, b)]
    assert un(orc.fq2_sub(A, B)) == [P.fq2_sub(x, y) for x, y in zip(a, b)]
    assert un(orc.fq2_inv(A)) == [P.fq2_inv(x) for x in a]


def test_c_oracle_domain_constants(orc):
    for log_d in (0, 1, 4, 10, 21, 23, 47):
        k = orc.domain_constants(log_d)
        d = 1 << log_d
        w = P.fr_root_of_unity(log_d)
        g = lambda name: P.fr_from_mont(orc.limbs_to_ints(k[name])[0])
        assert g("group_gen") == w and g("group_gen_inv") == pow(w, -1, P.R_MOD)
        assert g("size_inv") == pow(d, -1, P.R_MOD)
        assert g("generator") == 22 and g("generator_inv") == pow(22, -1, P.R_MOD)
        assert g("vanishing_inv") == pow(pow(22, d, P.R_MOD) - 1, -1, P.R_MOD)


def test_c_oracle_ntt_vs_definition(orc):
    # mirrors radix2/mod.rs:320-360 test_fft_correctness and :381-491 (fast == textbook incl. coset)
    rng = random.Random(3)
    for log_d in (0, 1, 2, 3, 5, 6):
        d = 1 << log_d
        for in_len in sorted({d, max(1, d - 3), (d + 1) // 2}):
            xs = _rand_fr(rng, in_len)
            X = orc.ints_to_limbs([P.fr_to_mont(x) for x in xs], 4)
            for kind, inv, coset in ((orc.FFT, False, False), (orc.IFFT, True, False),
                                     (orc.COSET_FFT, False, True), (orc.COSET_IFFT, True, True)):
                got = [P.fr_from_mont(v) for v in orc.limbs_to_ints(orc.ntt_fr(X, log_d, kind, in_len))]
                assert got == P.dft(xs, log_d, inverse=inv, coset=coset), (log_d, in_len, kind)
    # fft == Horner at w^i, coset_fft == Horner at 22 w^i  (size 2^7)
    log_d, d = 7, 128
    xs = _rand_fr(rng, d)
    X = orc.ints_to_limbs([P.fr_to_mont(x) for x in xs], 4)
    w = P.fr_root_of_unity(log_d)
    ev = [P.fr_from_mont(v) for v in orc.limbs_to_ints(orc.ntt_fr(X, log_d, orc.FFT))]
    cev = [P.fr_from_mont(v) for v in orc.limbs_to_ints(orc.ntt_fr(X, log_d, orc.COSET_FFT))]
    for i in (0, 1, 2, 63, 127):
        assert ev[i] == P.horner(xs, pow(w, i, P.R_MOD))
        assert cev[i] == P.horner(xs, 22 * pow(w, i, P.R_MOD) % P.R_MOD)
        xm = orc.ints_to_limbs([P.fr_to_mont(pow(w, i, P.R_MOD))], 4)
        assert P.fr_from_mont(orc.limbs_to_ints(orc.fr_horner(X, xm))[0]) == ev[i]
    # round trips (poly/src/test.rs:33-47)
    for k1, k2 in ((orc.FFT, orc.IFFT), (orc.COSET_FFT, orc.COSET_IFFT)):
        assert np.array_equal(orc.ntt_fr(orc.ntt_fr(X, log_d, k1), log_d, k2), X)


def _g1_mont(pt):
    return [P.fq_to_mont(pt[0]), P.fq_to_mont(pt[1])]


def _g2_mont(pt):
    return [P.fq_to_mont(pt[0][0]), P.fq_to_mont(pt[0][1]), P.fq_to_mont(pt[1][0]), P.fq_to_mont(pt[1][1])]


def _aff_from_limbs(orc, g, arr, is_inf):
    if is_inf:
        return P.INF
    v = [P.fq_from_mont(x) for x in orc.limbs_to_ints(arr.reshape(-1, 6))]
    return (v[0], v[1]) if g == 1 else ((v[0], v[1]), (v[2], v[3]))


def test_c_oracle_group_law_and_msm(orc):
    # mirrors algebra/test-templates/src/msm.rs:16-33 (Pippenger == naive, compared in affine), incl.
    # zero scalars, unit scalars, infinity bases, duplicate and opposite points.
    rng = random.Random(4)
    for g, F, gen, mont in ((1, P.F1, P.G1_GEN, _g1_mont), (2, P.F2, P.G2_GEN, _g2_mont)):
        n = 40 if g == 1 else 12
        pts = [P.ec_mul(F, rng.randrange(1, P.R_MOD), gen) for _ in range(n)]
        pts[3] = pts[2]                      # equal points -> doubling branch
        pts[5] = P.ec_neg(F, pts[4])         # opposite points -> infinity branch
        inf = np.zeros(n, dtype=np.uint8)
        inf[7] = 1                           # infinity base (flag set, coordinates arbitrary)
        eff = [P.INF if inf[i] else p for i, p in enumerate(pts)]
        bases = orc.ints_to_limbs([x for p in pts for x in mont(p)], 6).reshape(n, -1)
        sc = [rng.randrange(P.R_MOD) for _ in range(n)]
        sc[0], sc[1], sc[2], sc[3], sc[4], sc[5] = 0, 1, 5, 5, 9, 9
        S = orc.ints_to_limbs(sc, 4)
        for m in (n, 31, 33 if n >= 33 else n, 1, 0):       # c = 3 (<32) and c = log2*69/100+2 branches
            m = min(m, n)
            jac = orc.msm(g, bases[:m], inf[:m], S[:m])
            aff, is_inf = orc.jac_to_affine(g, jac)
            assert _aff_from_limbs(orc, g, aff, is_inf) == P.msm_naive(F, eff[:m], sc[:m]), (g, m)
        # AffineCurve::multi_scalar_mul: Montgomery scalars, |scalars| = |bases| + 1 (h vs h_query)
        SM = orc.fr_from_repr(np.vstack([S, orc.ints_to_limbs([123], 4)]))
        aff, is_inf = orc.jac_to_affine(g, orc.multi_scalar_mul(g, bases, inf, SM))
        assert _aff_from_limbs(orc, g, aff, is_inf) == P.msm_naive(F, eff, sc)
        # group law spot checks
        k = rng.randrange(P.R_MOD)
        jk = orc.scalar_mul(g, bases[0], False, orc.ints_to_limbs([k], 4))
        aff, is_inf = orc.jac_to_affine(g, jk)
        assert _aff_from_limbs(orc, g, aff, is_inf) == P.ec_mul(F, k, pts[0])
        j2 = orc.jac_add(g, jk, jk)
        aff2, i2 = orc.jac_to_affine(g, j2)
        assert _aff_from_limbs(orc, g, aff2, i2) == P.ec_mul(F, 2 * k % P.R_MOD, pts[0])
        aff3, i3 = orc.jac_to_affine(g, orc.jac_double(g, jk))
        assert np.array_equal(aff3, aff2) and i3 == i2
        jm = orc.jac_add_mixed(g, jk, bases[1])
        aff4, i4 = orc.jac_to_affine(g, jm)
        assert _aff_from_limbs(orc, g, aff4, i4) == P.ec_add(F, P.ec_mul(F, k, pts[0]), pts[1])


def test_window_rule():
    # variable_base.rs:21-25 with ark_std::log2 = ceil; values quoted in SURVEY.md row a10
    import math

    def c_of(size):
        lg = 0 if size == 0 else (size - 1).bit_length() if size & (size - 1) else size.bit_length() - 1
        return 3 if size < 32 else lg * 69 // 100 + 2
    assert c_of((1 << 20) + 1) == 16 and c_of((1 << 21) - 1) == 16 and c_of(1 << 20) == 15
    assert c_of((1 << 22) + 1) == 17 and c_of(1 << 18) == 14 and c_of(1 << 17) == 13 and c_of(11) == 3


def _random_csr(rng, m, n_vars, max_terms=5):
    row_ptr, col, coeff = [0], [], []
    for i in range(m):
        k = 0 if i % 7 == 3 else rng.randrange(1, max_terms + 1)   # some empty rows
        for _ in range(k):
            col.append(rng.randrange(n_vars))
            c = rng.choice([1, 1, P.R_MOD - 1, rng.randrange(P.R_MOD)])
            coeff.append(c)
        row_ptr.append(len(col))
    return row_ptr, col, coeff


def test_c_oracle_r1cs_matvec_vs_bigint(orc):
    """orc_r1cs_matvec (evaluate_constraint, r1cs_to_qap.rs:12-42) against plain modular arithmetic."""
    rng = random.Random(11)
    m, n_vars = 200, 64
    row_ptr, col, coeff = _random_csr(rng, m, n_vars)
    z = _rand_fr(rng, n_vars)
    want = [sum(coeff[t] * z[col[t]] for t in range(row_ptr[i], row_ptr[i + 1])) % P.R_MOD for i in range(m)]
    mont = lambda xs: orc.ints_to_limbs([x * P.FR_MONT_R % P.R_MOD for x in xs], 4)
    got = orc.r1cs_matvec(np.array(row_ptr, dtype=np.uint64), np.array(col, dtype=np.uint32), mont(coeff), mont(z))
    assert orc.limbs_to_ints(orc.fr_into_repr(got)) == want


def test_c_oracle_poly_div_linear_vs_bigint(orc):
    """orc_poly_div_linear (divide_with_q_and_r with divisor X - z) against Horner's rule on integers; also the
    defining identity p = q (X - z) + r and the leading-zero cases DensePolynomial truncates."""
    rng = random.Random(12)
    mont = lambda xs: orc.ints_to_limbs([x * P.FR_MONT_R % P.R_MOD for x in xs], 4)
    for n, tail_zeros in ((0, 0), (1, 0), (2, 0), (9, 0), (130, 0), (40, 3), (5, 5)):
        p = _rand_fr(rng, n)
        for k in range(tail_zeros):
            p[n - 1 - k] = 0
        z = rng.randrange(P.R_MOD)
        q_want = [0] * max(n - 1, 0)
        run = 0
        for i in range(n - 1, 0, -1):
            run = (p[i] + z * run) % P.R_MOD
            q_want[i - 1] = run
        r_want = (p[0] + z * run) % P.R_MOD if n else 0
        q, r = orc.poly_div_linear(mont(p) if n else np.zeros((0, 4), dtype=np.uint64), mont([z])[0])
        assert orc.limbs_to_ints(orc.fr_into_repr(q)) == q_want if n > 1 else len(q) == 0
        assert orc.limbs_to_ints(orc.fr_into_repr(r.reshape(1, 4))) == [r_want]
        # p == q * (X - z) + r coefficient-wise
        rec = [0] * n
        for i, c in enumerate(q_want):
            rec[i] = (rec[i] - z * c) % P.R_MOD
            rec[i + 1] = (rec[i + 1] + c) % P.R_MOD
        if n:
            rec[0] = (rec[0] + r_want) % P.R_MOD
        assert rec == p


def test_c_oracle_prefix_product_vs_bigint(orc):
    rng = random.Random(13)
    xs = _rand_fr(rng, 50)
    want, run = [], 1
    for v in xs:
        run = run * v % P.R_MOD
        want.append(run)
    got = orc.fr_prefix_product(orc.ints_to_limbs([x * P.FR_MONT_R % P.R_MOD for x in xs], 4))
    assert orc.limbs_to_ints(orc.fr_into_repr(got)) == want


def test_c_oracle_batch_inverse_vs_bigint(orc):
    """orc_fr_batch_inverse (serial_batch_inversion_and_mul, fields/mod.rs:642-677; the reference's own test is
    fields/mod.rs:704-727) against pow(x, -1, r), zeros skipped."""
    rng = random.Random(14)
    xs = _rand_fr(rng, 40)
    for i in (0, 7, 8, 39):
        xs[i] = 0
    coeff = rng.randrange(1, P.R_MOD)
    mont = lambda v: orc.ints_to_limbs([x * P.FR_MONT_R % P.R_MOD for x in v], 4)
    got = orc.fr_batch_inverse(mont(xs), mont([coeff])[0])
    assert orc.limbs_to_ints(orc.fr_into_repr(got)) == [coeff * pow(x, -1, P.R_MOD) % P.R_MOD if x else 0 for x in xs]

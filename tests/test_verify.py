"""The reference's own acceptance criterion for the whole path: the revealed proof VERIFIES (mpc-snarks/test.zsh runs groth16 / plonk / marlin and
asserts `verify_proof`, mpc-snarks/src/proof.rs:140-143).  The benchmark's proving key is synthetic (random points with known discrete logs), so its
proofs cannot verify; here the key is a REAL one -- generated, as groth16/src/generator.rs:60-230 does, from toxic waste (tau, alpha, beta, gamma, delta)
that the test knows -- handed to the prover as discrete logs (Groth16Local(key_scalars=...): the GPU builds the points).  The proof the hot path returns
(constraint evaluation, witness map with both opens, five MSMs on every share lane, create_proof's group steps; parties' shares added up) is then checked

  (1) element by element against [a] G1, [b] G2, [c] G1 for the exponents a, b, c that the Groth16 prover equations give (groth16/src/prover.rs:110-178)
      from the plain witness, r, s and the quotient h -- points made by the CPU checker's double-and-add, compared in affine; and
  (2) against the verification equation e(A, B) = e(alpha, beta) e(sum_i x_i gamma_abc_i, gamma) e(C, delta) (groth16/src/verifier.rs:40-62), which with
      every discrete log known is a b = alpha beta + sum_i x_i (beta u_i + alpha v_i + w_i) + c delta in Fr -- no pairing needed.

(2) holds only if h is the true quotient (A B - C) / Z of the QAP over the reference's domain (omega = LARGE^3 squared down, fields/mod.rs:360-367): a wrong
root of unity, coset, 1/D or vanishing-polynomial factor anywhere in the witness map breaks it, whatever the checker's restatement says.  Plain big-integer
Python; nothing here comes from the library except the proof."""
import numpy as np
import pytest

from groth16_real_key import R_INV, expected_exponents, key_scalars, limbs, real_key
from util import R_MOD, dot_mod_r, ints_to_limbs, limbs_to_ints, rand_fr_canonical

pytestmark = pytest.mark.gpu


def prove_and_verify(orc, N, parties, scheme, **prover_kw):
    import torch
    import czk_amd as czk
    from czk_amd.provers import Groth16Local
    key = real_key(N, limbs_to_ints(rand_fr_canonical(0x7A11 + N, 5)))
    D = key["D"]
    # the domain generator derived from the reference's constants is the checker's (and, through the parity tests, the library's)
    assert key["omega"] == limbs_to_ints(orc.domain_constants(key["log_d"])["group_gen"].reshape(1, 4))[0] * R_INV % R_MOD
    ks = key_scalars(key)
    rs = rand_fr_canonical(0xC0FFEE + 77 + N, 2)
    r, s = limbs_to_ints(rs)
    ninv = pow(parties, -1, R_MOD) if scheme == "gsz" else 1       # Shamir shares on the n-th roots of unity: p(0) = (1 / n) sum_j p(w^j)
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        ctx = czk.Context(0, ts.cuda_stream)
        p = Groth16Local(czk, ctx, N, parties, scheme=scheme, key_scalars=ks, **prover_kw)
        p.step()
        torch.cuda.synchronize()
        proof = p.create_proof({k: v.copy() for k, v in p.results.items()}, rs[0], rs[1])      # (expands one-MSM-per-party results to the lanes)
        lpp = p.lpp
        opened = {}                                                # open: the parties' sh lanes added up
        for k, g in (("a", 1), ("b", 2), ("c", 1)):
            acc = proof[k][0]
            for j in range(1, parties):
                acc = ctx.jac_add(g, acc, proof[k][lpp * j])
            if scheme == "gsz":
                acc = ctx.jac_scalar_mul(g, acc, ints_to_limbs([ninv], 4)[0])
            opened[k] = ctx.jac_to_affine(g, acc)
        h_lanes = p.ab.cpu().numpy().view(np.uint64)
        mac_bad = int(torch.count_nonzero(p.chk).item()) if scheme == "spdz" else 0
        del p
        ctx.close()
    assert mac_bad == 0
    sh_lanes = [h_lanes[lpp * j] for j in range(parties)]          # shares of h (Montgomery limbs: x R^-1)
    assert sum(limbs_to_ints(ln[D - 1:D])[0] for ln in sh_lanes) % R_MOD == 0            # deg h <= D - 2
    h_acc = sum(dot_mod_r(ln[:D - 1], ks["h"]) for ln in sh_lanes) * R_INV % R_MOD * ninv % R_MOD
    w0 = limbs_to_ints(rand_fr_canonical(0xC0FFEE, 1))[0]
    a_exp, b_exp, c_exp, verifies, qap = expected_exponents(key, w0, r, s, h_acc)
    # (1) the proof's elements are [a] G1, [b] G2, [c] G1 (the checker's double-and-add on the generators)
    for k, g, e in (("a", 1, a_exp), ("b", 2, b_exp), ("c", 1, c_exp)):
        want, winf = orc.jac_to_affine(g, orc.scalar_mul(g, orc.generator_affine(g), 0, ints_to_limbs([e], 4)[0]))
        got, ginf = opened[k]
        assert not ginf[0] and not winf and np.array_equal(np.ravel(got[0]), np.ravel(want)), (k, N, parties, scheme)
    # (2) the verification equation, and the QAP identity behind it on its own
    assert verifies, "the proof does not verify"
    assert qap


@pytest.mark.parametrize("n_constraints,parties,scheme", [(10, 2, "spdz"), (1000, 2, "spdz"), (1000, 3, "gsz"), (333, 1, "hbc"), (4094, 2, "hbc")])
def test_groth16_proof_verifies_under_a_real_key(orc, n_constraints, parties, scheme):
    prove_and_verify(orc, n_constraints, parties, scheme)


@pytest.mark.parametrize("kw", [{"no_tables": True}, {"mac_msm_from_sh": True}])
def test_groth16_proof_verifies_in_the_other_msm_forms(orc, kw):
    """the key registered without window tables (CZK_MEM_NO_TABLES: one bucket set per window), and the reference-shaped SPDZ form (one MSM per party over its sh
    lane, the result used for the sh and the mac group share: spdz.rs:440-446)"""
    prove_and_verify(orc, 3000, 2, "spdz", **kw)


def test_groth16_full_size_proof_verifies(orc):
    """BASELINE configs[1] itself -- 2^20 constraints, SPDZ, two parties, four share lanes on one GPU -- under a real key: the proof verifies."""
    prove_and_verify(orc, 1 << 20, 2, "spdz")


# ---------------------------------------------------------------------------------------------------------------------------------------
# Plonk: a SATISFIED circuit, the reference's verifier (mpc-plonk/src/lib.rs:451-590)
# ---------------------------------------------------------------------------------------------------------------------------------------
def satisfied_plonk_inputs(B, polyvm, n_gates: int, seed: int):
    """A circuit layout (mpc-plonk/src/relations/flat.rs) that IS satisfied, unlike the benchmark's random polynomials: a chain of n_gates gates alternating
    v -> v * v (s = 0) and v -> v + v (s = 1); gate i's wires sit at w^(3i), w^(3i+1), w^(3i+2) of the wire domain (left, right, out), so that
    s (p + p(wX)) + (1 - s) p p(wX) - p(w^2 X) vanishes on the gate domain <w^3>; the copy constraints out_{i-1} = left_i = right_i are the cycles of the
    wiring permutation, given as the polynomial w(X) with w(w^j) = w^sigma(j).  Returns (inputs for plonk_prove, the wire values, the generator w)."""
    G, W = n_gates, 3 * n_gates
    w = B.root_of_unity(W)
    assert pow(w, W, R_MOD) == 1 and pow(w, 3 * (G // 2), R_MOD) != 1
    v = limbs_to_ints(rand_fr_canonical(seed, 1))[0]
    e, s_ev = [], []
    for i in range(G):
        out = v * v % R_MOD if i % 2 == 0 else 2 * v % R_MOD
        e += [v, v, out]
        s_ev.append(i % 2)
        v = out
    sigma = list(range(W))
    sigma[0], sigma[1] = 1, 0
    for i in range(1, G):
        a, b, c = 3 * (i - 1) + 2, 3 * i, 3 * i + 1
        sigma[a], sigma[b], sigma[c] = b, c, a
    assert all(e[j] == e[sigma[j]] for j in range(W))
    rr = (1 << 256) % R_MOD
    pub = lambda vals: B.upload(limbs([x * rr % R_MOD for x in vals])[None])   # noqa: E731  (1, n, 4) public array, Montgomery form
    p_pub = B.ntt(pub(e), W, polyvm.IFFT)
    wp, acc = [], 1
    for _ in range(W):
        wp.append(acc)
        acc = acc * w % R_MOD
    inp = {"n_gates": G, "p": polyvm.shared_copy(B, p_pub), "s": B.ntt(pub(s_ev), G, polyvm.IFFT), "w": B.ntt(pub([wp[sg] for sg in sigma]), W, polyvm.IFFT)}
    return inp, e, w


@pytest.mark.parametrize("n_gates,parties", [(8, 3), (256, 3), (2048, 2), (1 << 18, 3)])   # the last: BASELINE configs[2]'s size
def test_plonk_proof_of_a_satisfied_circuit_verifies(n_gates, parties):
    """mpc-plonk's Prover::prove on the GPU path for a satisfied circuit, then the reference's Verifier::verify (lib.rs:511-590) on what it returns: every
    KZG opening against its commitment (with the synthetic SRS's known tau: C - [v] G == [tau - x] W, bench.verify_openings) and the four identities on the
    opened values -- public wire, gates, unit product (partial products and t(w^(k-1)) = 1), wiring."""
    import czk_amd
    from czk_amd import polyvm
    import bench
    ctx = polyvm.shared_stream_context(czk_amd)
    B = polyvm.GpuBackend(czk_amd, ctx, parties, polyvm.plonk_max_degree(n_gates))
    inp, e, w = satisfied_plonk_inputs(B, polyvm, n_gates, 0x51A7 + n_gates)
    G, W = n_gates, 3 * n_gates
    vk = {"s_cmt": B.commit(inp["s"]), "w_cmt": B.commit(inp["w"])}            # the index polynomials' commitments (the verifying key)
    out = polyvm.plonk_prove(B, inp)
    out.update(polyvm.resolved(vk))
    out["gates_s_open"]["of"], out["w_x_open"]["of"] = "s", "w"
    n_open = sum(1 for o in out.values() if isinstance(o, dict) and o.get("of"))
    chk = bench.verify_openings(czk_amd, ctx, B, out)
    assert chk["results_checked"] and chk["results_checked_points"] == parties * (n_open - 2) + 2

    def val(label):                                                            # the revealed evaluation: the king_share stand-in puts the value on every lane
        vs = [polyvm.unmont(x) for x in out[label]["value"]]
        assert all(x == vs[0] for x in vs), label
        return vs[0]
    ch = polyvm.challenge
    # verify_public (:526-540): one public wire at w^1 with value e[1]
    x = ch("plonk.public.x")
    assert (val("pub_p_open") - e[1]) % R_MOD == val("pub_q_open") * (x - w) % R_MOD
    # verify_gates (:542-560)
    x = ch("plonk.gates.x")
    s, q, p, pw, pww = (val(k) for k in ("gates_s_open", "gates_q_open", "gates_p_open", "gates_p_w_open", "gates_p_w2_open"))
    assert (s * (p + pw) + (1 - s) * p * pw - pww) % R_MOD == q * polyvm.vanishing(G, x) % R_MOD
    # verify_unit_product (:451-474)
    r = ch("plonk.product.r")
    assert (val("t_wr_open") - val("t_r_open") * val("f_wr_open")) % R_MOD == polyvm.vanishing(W, r) * val("q_r_open") % R_MOD
    assert val("t_wk_open") == 1
    # verify_wiring (:561-582)
    y, z, x = ch("plonk.wiring.y"), ch("plonk.wiring.z"), ch("plonk.wiring.x")
    p_x, l1_x, w_x, l2 = val("p_x_open"), val("l1_x_open"), val("w_x_open"), val("l2_q_x_open")
    assert ((p_x + y * x + z) * l1_x - (p_x + y * w_x + z)) % R_MOD == l2 * polyvm.vanishing(W, x) % R_MOD
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------------------------
# Marlin: a real index of a satisfied R1CS instance, the AHP verifier's decision (marlin/src/ahp/mod.rs:115-260, marlin/src/lib.rs:320-450)
# ---------------------------------------------------------------------------------------------------------------------------------------
def marlin_real_inputs(B, polyvm, H: int, seed: int):
    """The benchmark's Marlin inputs are work-shaped stand-ins (random index data, random z_a / z_b); this builds the real thing for a satisfied instance:
    |H| constraints and variables -- formatted input [1, out], a squaring chain w_{i+1} = w_i^2 ending in `out`, and two rows that use the input columns
    (1 * out = out, 3 * 1 = 3) -- arithmetised as the indexer does (ahp/constraint_systems.rs:152-260: row / col / val of M* over K with val divided by
    u_H(col, col), variables placed on H by reindex_by_subdomain), with z_a = A z, z_b = B z, a mask polynomial that sums to zero over H (prover.rs:376-380)
    and the transposed matrices calculate_t walks (prover.rs:400-416)."""
    X, K = 2, H
    ratio = H // X
    b_size = polyvm.next_pow2(3 * K - 3)
    h = B.root_of_unity(H)
    elems, acc = [], 1
    for _ in range(H):
        elems.append(acc)
        acc = acc * h % R_MOD
    reindex = lambda i: i * ratio if i < X else (i - X) + (i - X) // (ratio - 1) + 1   # noqa: E731  (domain/mod.rs:196-218)
    nw = H - X
    w = [limbs_to_ints(rand_fr_canonical(seed, 1))[0]]
    for _ in range(nw - 1):
        w.append(w[-1] * w[-1] % R_MOD)
    out_v = w[-1] * w[-1] % R_MOD
    z = [1, out_v] + w                                           # variable i: inputs first, then the witness
    var_w = lambda i: X + i                                      # noqa: E731
    rows = {"a": [], "b": [], "c": []}
    for i in range(nw):
        rows["a"].append([(1, var_w(i))])
        rows["b"].append([(1, var_w(i))])
        rows["c"].append([(1, var_w(i + 1) if i + 1 < nw else 1)])
    rows["a"] += [[(1, 0)], [(3, 0)]]
    rows["b"] += [[(1, 1)], [(1, 0)]]
    rows["c"] += [[(1, 1)], [(3, 0)]]
    mz = {m: [sum(cf * z[i] for cf, i in row) % R_MOD for row in rows[m]] for m in "abc"}
    assert all(a * b % R_MOD == c for a, b, c in zip(mz["a"], mz["b"], mz["c"])) and len(rows["a"]) == H
    rr = (1 << 256) % R_MOD
    pub = lambda vals: B.upload(limbs([x * rr % R_MOD for x in vals])[None])   # noqa: E731
    h_inv = pow(H, -1, R_MOD)                                                    # 1 / u_H(e, e) = 1 / (|H| e^(|H| - 1)) = e / |H| on H
    inp = {"H": H, "K": K, "X": X, "b_size": b_size, "star": {}, "matrices_T": {}, "real_lcs": True, "t_rows": None}
    index_polys = []
    for m in "abc":
        row_v, col_v, val_v, trans = [], [], [], [[] for _ in range(H)]
        for r, row in enumerate(rows[m]):
            for cf, i in sorted(row, key=lambda t: t[1]):
                cv = elems[reindex(i)]
                row_v.append(cv)                                 # "we are dealing with the transpose of M" (:191-194)
                col_v.append(elems[r])
                val_v.append(cf * cv % R_MOD * h_inv % R_MOD)
                trans[reindex(i)].append((r, cf))
        pad = K - len(row_v)
        row_v += [elems[0]] * pad
        col_v += [elems[0]] * pad
        val_v += [0] * pad
        rc_v = [a * b % R_MOD for a, b in zip(row_v, col_v)]
        on_k = [pub(v) for v in (row_v, col_v, val_v)]
        polys = [B.ntt(pub(v), K, polyvm.IFFT) for v in (row_v, col_v, val_v, rc_v)]           # row, col, val, row_col
        on_b = [B.ntt(polys[j], b_size, polyvm.FFT) for j in (0, 1, 3, 2)]                     # row, col, row_col, val on B
        inp["star"][m] = {"on_K": on_k, "on_B": on_b}
        index_polys += polys
        rp, cols, cfs = [0], [], []
        for p in range(H):
            for r, cf in trans[p]:
                cols.append(r)
                cfs.append(cf * rr % R_MOD)
            rp.append(len(cols))
        inp["matrices_T"][m] = B.matrix(np.array(rp, dtype=np.uint64), np.array(cols, dtype=np.uint32), limbs(cfs), H)
    inp["index_polys"] = index_polys
    cm = [B.commit(a) for a in index_polys]
    B.transcript_point()
    inp["index_cmts"] = polyvm.resolved(cm)
    inp["x"] = pub([1, out_v])
    w_full = [z[k // ratio] if k % ratio == 0 else w[k - k // ratio - 1] for k in range(H)]
    assert all(w_full[reindex(i)] == z[i] for i in range(H))
    inp["w"] = polyvm.shared_copy(B, pub(w_full))
    inp["z_a"] = polyvm.shared_copy(B, pub(mz["a"]))
    inp["z_b"] = polyvm.shared_copy(B, pub(mz["b"]))
    mask = limbs_to_ints(rand_fr_canonical(seed + 5, 3 * H))
    mask[0] = (mask[0] - sum(mask[j] for j in range(0, 3 * H, H))) % R_MOD     # the remainder mod v_H has constant term 0: the mask sums to zero over H
    inp["mask_poly"] = polyvm.shared_copy(B, pub(mask))
    return inp


@pytest.mark.parametrize("H", [8, 64, 1024, 1 << 16])   # (|H| = 2^20, BASELINE configs[3]'s size: 27 s, run by hand -- profiles/r05_verification_runs.txt)
def test_marlin_proof_of_a_satisfied_instance_verifies(H):
    """Marlin's AHP prover rounds, commitments and batched openings on the GPU path for a REAL index and a satisfied instance (the opt-in paths of
    polyvm.marlin_prove: calculate_t over the transposed matrices, the linear combinations' real coefficients), then the verifier: every KZG opening -- the
    two batched ones against the folded commitments, the degree-bound witnesses -- with the known tau (bench.verify_openings), and the AHP decision: the outer
    sumcheck combination evaluates to zero at beta and the inner one at gamma (ahp/mod.rs:168-183, 233-250), constants included."""
    import czk_amd
    from czk_amd import polyvm
    import bench
    ctx = polyvm.shared_stream_context(czk_amd)
    lanes = 2
    B = polyvm.GpuBackend(czk_amd, ctx, lanes, polyvm.marlin_max_degree(H), lift=(1,) * lanes)    # public data on every lane: each lane is the plain prover
    inp = marlin_real_inputs(B, polyvm, H, 0x3A21 + H)
    out = polyvm.marlin_prove(B, inp)
    chk = bench.verify_openings(czk_amd, ctx, B, out)
    assert chk["results_checked"] and chk["results_checked_points"] == 2 * lanes + 2      # beta: share lanes; gamma: index polynomials, public
    # the evaluations the prover publicized, by (polynomial, point): replay of marlin_prove's lc_eval order
    lcs, consts = out["lcs"], out["lc_consts"]
    query_beta = ("g_1", "outer_sumcheck", "t", "z_b")
    it = {"beta": iter(out["evals_beta"]), "gamma": iter(out["evals_gamma"])}
    ev = {}

    def take(label, tag):
        for _, name in lcs[label]:
            v = [polyvm.unmont(x) for x in next(it[tag])]
            assert all(x == v[0] for x in v), (label, name)
            assert ev.setdefault((name, tag), v[0]) == v[0]
    for label, tag in (("z_b", "beta"), ("t", "beta"), ("g_1", "beta"), ("a_denom", "gamma"), ("b_denom", "gamma"), ("c_denom", "gamma"), ("g_2", "gamma")):
        take(label, tag)
    for label in sorted(lcs):
        take(label, "beta" if label in query_beta else "gamma")
    assert next(it["beta"], None) is None and next(it["gamma"], None) is None
    lc_at = lambda label, tag: (sum(cf * ev[(name, tag)] for cf, name in lcs[label]) + consts.get(label, 0)) % R_MOD   # noqa: E731
    assert lc_at("outer_sumcheck", "beta") == 0, "outer sumcheck"
    assert lc_at("inner_sumcheck", "gamma") == 0, "inner sumcheck"
    # the batched openings opened what the verifier folds: sum_j ch^j (LC_j without its constant), a degree-bounded polynomial taking two challenges
    ch = polyvm.challenge("marlin.opening_challenge")
    for tag, labels in (("beta", query_beta), ("gamma", ("a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"))):
        want, c = 0, 1
        for label in labels:
            want = (want + c * (lc_at(label, tag) - consts.get(label, 0))) % R_MOD
            c = c * ch % R_MOD * (ch if label in ("g_1", "g_2") else 1) % R_MOD
        got = [polyvm.unmont(x) for x in out["open_" + tag]["value"]]
        assert all(g == want for g in got), tag
    ctx.close()

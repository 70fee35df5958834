"""Satisfied Plonk circuits and real Marlin indices for the polynomial provers of collaborative-zksnark_amd/polyvm.py, and the reference verifiers' equations on
what the provers return -- plain big-integer Python over a polyvm Backend (test infrastructure: tests/test_verify.py, bench.py's `proof_verifies` legs).
The benchmark inputs of both provers are work-shaped stand-ins (random polynomials / random index data); these are the real thing."""
import numpy as np

from groth16_real_key import R_MOD, limbs

_RR = (1 << 256) % R_MOD


def _pub(B, vals):
    """(1, n, 4) public array of canonical integers, Montgomery form"""
    return B.upload(limbs([x * _RR % R_MOD for x in vals])[None])


def _rand_ints(seed, n):
    from czk_amd.provers import rand_fr_canonical
    a = rand_fr_canonical(seed, n)
    return [sum(int(v[j]) << (64 * j) for j in range(4)) for v in a]


def satisfied_plonk_inputs(B, polyvm, n_gates: int, seed: int):
    """A circuit layout (mpc-plonk/src/relations/flat.rs) that IS satisfied, unlike the benchmark's random polynomials: a chain of n_gates gates alternating
    v -> v * v (s = 0) and v -> v + v (s = 1); gate i's wires sit at w^(3i), w^(3i+1), w^(3i+2) of the wire domain (left, right, out), so that
    s (p + p(wX)) + (1 - s) p p(wX) - p(w^2 X) vanishes on the gate domain <w^3>; the copy constraints out_{i-1} = left_i = right_i are the cycles of the
    wiring permutation, given as the polynomial w(X) with w(w^j) = w^sigma(j).  Returns (inputs for plonk_prove, the wire values, the generator w)."""
    G, W = n_gates, 3 * n_gates
    w = B.root_of_unity(W)
    assert pow(w, W, R_MOD) == 1 and pow(w, 3 * (G // 2), R_MOD) != 1
    v = _rand_ints(seed, 1)[0]
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
    pub = lambda vals: _pub(B, vals)   # noqa: E731
    p_pub = B.ntt(pub(e), W, polyvm.IFFT)
    wp, acc = [], 1
    for _ in range(W):
        wp.append(acc)
        acc = acc * w % R_MOD
    inp = {"n_gates": G, "p": polyvm.shared_copy(B, p_pub), "s": B.ntt(pub(s_ev), G, polyvm.IFFT), "w": B.ntt(pub([wp[sg] for sg in sigma]), W, polyvm.IFFT)}
    return inp, e, w


def plonk_verify(polyvm, out, e, w, n_gates: int):
    """The four identities of the reference's Verifier::verify (mpc-plonk/src/lib.rs:451-590) on the opened values of plonk_prove's output (the KZG openings
    themselves are checked by bench.verify_openings); e = the circuit's wire values, w = the wire domain's generator.  Raises AssertionError."""
    G, W = n_gates, 3 * n_gates

    def val(label):                                                            # the revealed evaluation: the king_share stand-in puts the value on every lane
        vs = [polyvm.unmont(x) for x in out[label]["value"]]
        assert all(x == vs[0] for x in vs), label
        return vs[0]
    ch = polyvm.challenge
    # verify_public (:526-540): one public wire at w^1 with value e[1]
    x = ch("plonk.public.x")
    assert (val("pub_p_open") - e[1]) % R_MOD == val("pub_q_open") * (x - w) % R_MOD, "public wire"
    # verify_gates (:542-560)
    x = ch("plonk.gates.x")
    s, q, p, pw, pww = (val(k) for k in ("gates_s_open", "gates_q_open", "gates_p_open", "gates_p_w_open", "gates_p_w2_open"))
    assert (s * (p + pw) + (1 - s) * p * pw - pww) % R_MOD == q * polyvm.vanishing(G, x) % R_MOD, "gates"
    # verify_unit_product (:451-474)
    r = ch("plonk.product.r")
    assert (val("t_wr_open") - val("t_r_open") * val("f_wr_open")) % R_MOD == polyvm.vanishing(W, r) * val("q_r_open") % R_MOD, "partial products"
    assert val("t_wk_open") == 1, "total product"
    # verify_wiring (:561-582)
    y, z, x = ch("plonk.wiring.y"), ch("plonk.wiring.z"), ch("plonk.wiring.x")
    p_x, l1_x, w_x, l2 = val("p_x_open"), val("l1_x_open"), val("w_x_open"), val("l2_q_x_open")
    assert ((p_x + y * x + z) * l1_x - (p_x + y * w_x + z)) % R_MOD == l2 * polyvm.vanishing(W, x) % R_MOD, "wiring"


def plonk_prove_and_verify(polyvm, B, verify_openings, n_gates: int, seed: int = 0x51A7):
    """plonk_prove on a satisfied circuit of n_gates gates on backend B, then the verifier; returns the number of KZG openings checked"""
    inp, e, w = satisfied_plonk_inputs(B, polyvm, n_gates, seed + n_gates)
    vk = {"s_cmt": B.commit(inp["s"]), "w_cmt": B.commit(inp["w"])}            # the index polynomials' commitments (the verifying key)
    out = polyvm.plonk_prove(B, inp)
    out.update(polyvm.resolved(vk))
    out["gates_s_open"]["of"], out["w_x_open"]["of"] = "s", "w"
    n_open = sum(1 for o in out.values() if isinstance(o, dict) and o.get("of"))
    chk = verify_openings(out)
    assert chk["results_checked"] and chk["results_checked_points"] == B.lanes * (n_open - 2) + 2
    plonk_verify(polyvm, out, e, w, n_gates)
    return chk["results_checked_points"]


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
    w = _rand_ints(seed, 1)
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
    rr = _RR
    pub = lambda vals: _pub(B, vals)   # noqa: E731
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
    inp["x_ints"] = [1, out_v]
    w_full = [z[k // ratio] if k % ratio == 0 else w[k - k // ratio - 1] for k in range(H)]
    assert all(w_full[reindex(i)] == z[i] for i in range(H))
    inp["w"] = polyvm.shared_copy(B, pub(w_full))
    inp["z_a"] = polyvm.shared_copy(B, pub(mz["a"]))
    inp["z_b"] = polyvm.shared_copy(B, pub(mz["b"]))
    mask = _rand_ints(seed + 5, 3 * H)
    mask[0] = (mask[0] - sum(mask[j] for j in range(0, 3 * H, H))) % R_MOD     # the remainder mod v_H has constant term 0: the mask sums to zero over H
    inp["mask_poly"] = polyvm.shared_copy(B, pub(mask))
    return inp


def marlin_verify(polyvm, out):
    """The AHP verifier's decision on marlin_prove's output for a real index (out["lcs"], out["lc_consts"] present): the outer sumcheck combination is zero at
    beta, the inner one at gamma (ahp/mod.rs:168-183, 233-250), constants included, and the batched openings opened what the verifier folds.  The KZG openings
    themselves are checked by bench.verify_openings.  Raises AssertionError."""
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


def marlin_prove_and_verify(polyvm, B, verify_openings, H: int, seed: int = 0x3A21):
    """marlin_prove on a real index of a satisfied instance with |H| = H on backend B (public data lifted onto every lane), then the verifier"""
    assert all(B.lift)
    inp = marlin_real_inputs(B, polyvm, H, seed + H)
    out = polyvm.marlin_prove(B, inp)
    chk = verify_openings(out)
    assert chk["results_checked"] and chk["results_checked_points"] == 2 * B.lanes + 2      # beta: share lanes; gamma: index polynomials, public
    marlin_verify(polyvm, out)
    return chk["results_checked_points"]

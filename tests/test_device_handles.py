"""Device-resident share lanes (czk_lanes_*) and the torch-free C++ host built on them (tools/host_demo.cpp `bench`,
tools/groth16_host.hpp over include/czk.hpp): the reference-side binding's route to the headline.

The reference keeps its share vectors alive across the witness map and feeds `h` straight into the MSM
(mpc-snarks/src/groth/r1cs_to_qap.rs:85-110, mpc-snarks/src/groth/prover.rs:104); the GPU tests here check that a caller holding
only C-ABI handles gets the checker's values: h bit-exact on every share lane, the five MSMs of every lane equal in affine."""
import os
import subprocess

import numpy as np
import pytest

from util import rand_fr_canonical

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041


# ---------------------------------------------------------------------------------------------------------------
# the benchmark's inputs, rebuilt on the host from the seeds (collaborative-zksnark_amd/provers.py, tools/groth16_host.hpp)
# ---------------------------------------------------------------------------------------------------------------
def groth16_inputs(orc, N, parties=2, seed=0xC0FFEE):
    """Share lanes of the squaring circuit (mpc-snarks/src/proof.rs:304-344): lane 2 j + m = party j, m = 0 sh / 1 mac (mac key 1).
    Returns a0, b0, c0 (L, D, 4) constraint evaluations, wit (L, N, 4), asg (L, N + 1, 4), log_d."""
    w0 = rand_fr_canonical(seed, 1)[0]
    v = sum(int(w0[j]) << (64 * j) for j in range(4))
    chain = [v]
    for _ in range(N):
        chain.append(chain[-1] * chain[-1] % R_MOD)
    R = (1 << 256) % R_MOD
    wm = np.frombuffer(b"".join(((c * R) % R_MOD).to_bytes(32, "little") for c in chain), dtype=np.uint64).reshape(-1, 4).copy()
    sh, rest = [], wm
    for p in range(parties - 1):
        rm = orc.fr_from_repr(rand_fr_canonical(seed + 17 * (p + 1), N + 1))
        rest = orc.fr_sub(rest, rm)
        sh.append(rm)
    sh.append(rest)
    one = wm[:1] * 0
    one[0] = np.frombuffer((R % R_MOD).to_bytes(32, "little"), dtype=np.uint64)
    log_d = (N + 1).bit_length()
    D, L = 1 << log_d, 2 * parties
    a0, b0, c0 = (np.zeros((L, D, 4), dtype=np.uint64) for _ in range(3))
    wit, asg = np.zeros((L, N, 4), dtype=np.uint64), np.zeros((L, N + 1, 4), dtype=np.uint64)
    for j in range(parties):
        for m in range(2):
            ln = 2 * j + m
            a0[ln, :N] = sh[j][:N]
            b0[ln, :N] = sh[j][:N]
            c0[ln, :N] = sh[j][1:N + 1]
            if j == 0:
                a0[ln, N] = one[0]
            a0[ln, N + 1] = sh[j][N]
            wit[ln] = sh[j][:N]
            asg[ln, 0] = sh[j][N]
            asg[ln, 1:] = sh[j][:N]
    return a0, b0, c0, wit, asg, log_d, one[0]


def beaver_explicit_h(orc, a0, b0, c0, log_d, one):
    """h of every lane by the reference's sequence: witness_map (r1cs_to_qap.rs:85-110) with batch_product_in_place as the Beaver
    protocol on the stand-in triples -- king (1, 1, 1), others (0, 0, 0) (wire/field.rs:41-60, share/field.rs:97-127)."""
    L, D = a0.shape[0], a0.shape[1]
    ones = np.tile(one, (D, 1))
    zeros = np.zeros((D, 4), dtype=np.uint64)
    t = [ones if ln < 2 else zeros for ln in range(L)]            # tx = ty = tz per lane
    A = [orc.witness_map_pre(a0[ln], b0[ln], log_d) for ln in range(L)]
    sa = [orc.fr_add(A[ln][0], t[ln]) for ln in range(L)]
    sb = [orc.fr_add(A[ln][1], t[ln]) for ln in range(L)]
    sx, oy = sa[0], sb[0]
    for p in range(1, L // 2):                                     # open = sum of the parties' sh lanes
        sx, oy = orc.fr_add(sx, sa[2 * p]), orc.fr_add(oy, sb[2 * p])
    hs = []
    for ln in range(L):
        ab = orc.fr_sub(orc.fr_sub(t[ln], orc.fr_mul(t[ln], sx)), orc.fr_mul(t[ln], oy))
        if ln < 2:
            ab = orc.fr_add(ab, orc.fr_mul(sx, oy))                # the king applies the shift on both of its lanes
        hs.append(orc.witness_map_post(ab, c0[ln], log_d))
    return hs


def beaver_shortcut_lanes(orc, a0, b0, c0):
    """Inputs (a', b', c') whose PLAIN per-lane witness map equals the Beaver sequence above on the stand-in triples: on the
    king's lanes z - y sx - x oy + sx oy = (sx - 1)(oy - 1) = (sum_p a_p)(sum_p b_p), on every other lane the combination is 0.
    The transforms are linear, so a' = sum of the sh lanes, b' likewise (king); a' = b' = 0 (others); c' = c."""
    L = a0.shape[0]
    at, bt = a0[0].copy(), b0[0].copy()
    for p in range(1, L // 2):
        at, bt = orc.fr_add(at, a0[2 * p]), orc.fr_add(bt, b0[2 * p])
    a1, b1 = np.zeros_like(a0), np.zeros_like(b0)
    a1[0] = a1[1] = at
    b1[0] = b1[1] = bt
    return a1, b1, c0.copy()


def groth16_pk_extras(orc_or_ctx_points):
    """the synthetic key's remaining elements (provers.py / groth16_host.hpp): G1 [alpha_g1, beta_g1, delta_g1, a_query0] = [k] G for
    k = rand_fr_canonical(0xBA5E5 + 6, 4), G2 [beta_g2, delta_g2] for rand_fr_canonical(0xBA5E5 + 7, 2); `orc_or_ctx_points(g, k)` -> affine"""
    g1 = orc_or_ctx_points(1, rand_fr_canonical(0xBA5E5 + 6, 4))
    g2 = orc_or_ctx_points(2, rand_fr_canonical(0xBA5E5 + 7, 2))
    return {"alpha_g1": g1[0], "beta_g1": g1[1], "delta_g1": g1[2], "a_query0": g1[3], "beta_g2": g2[0], "delta_g2": g2[1]}


def checker_create_proof(orc, pk, res, r, s, king_lanes):
    """create_proof after its MSMs (mpc-snarks/src/groth/prover.rs:110-178) and calculate_coeff (:216-232) restated with the CHECKER's group
    law, lane by lane, for public r, s (canonical limbs).  res: {"h","l","a","b_g1","b_g2"} -> per-lane Jacobian limbs.  Public group elements
    are added on the king's lanes only (Public + Shared = shift).  Returns per lane {"a","b","c"} -> (affine limbs, infinity flag)."""
    from util import ints_to_limbs
    neg_one = ints_to_limbs([R_MOD - 1], 4)[0]

    def mul(g, jac, k):                                            # ProjectiveCurve::mul (algebra/ec/src/lib.rs:215-226) through the affine form
        aff, inf = orc.jac_to_affine(g, jac)
        return orc.scalar_mul(g, aff, bool(inf), k)

    def proj(g, aff):                                              # into_projective
        return orc.scalar_mul(g, aff, False, ints_to_limbs([1], 4)[0])
    delta_g1, delta_g2 = proj(1, pk["delta_g1"]), proj(2, pk["delta_g2"])
    r_s_delta_g1 = mul(1, mul(1, delta_g1, r), s)
    r_g1, s_g1, s_g2 = mul(1, delta_g1, r), mul(1, delta_g1, s), mul(2, delta_g2, s)

    def calculate_coeff(g, initial, el, el_inf, acc, vk_param, king):
        if not king:
            return acc
        out = orc.jac_add_mixed(g, initial, el, el_inf)            # res = initial; res.add_assign_mixed(&el)
        out = orc.jac_add(g, out, acc)                              # res += &acc
        return orc.jac_add_mixed(g, out, vk_param, False)          # res.add_assign_mixed(&vk_param)
    out = []
    for ln in range(len(res["h"])):
        king = ln in king_lanes
        g_a = calculate_coeff(1, r_g1, pk["a_query0"], False, res["a"][ln], pk["alpha_g1"], king)
        s_g_a = mul(1, g_a, s)
        g1_b = calculate_coeff(1, s_g1, np.zeros(12, np.uint64), True, res["b_g1"][ln], pk["beta_g1"], king)
        g2_b = calculate_coeff(2, s_g2, np.zeros(24, np.uint64), True, res["b_g2"][ln], pk["beta_g2"], king)
        r_g1_b = mul(1, g1_b, r)
        g_c = orc.jac_add(1, s_g_a, r_g1_b)
        if king:
            g_c = orc.jac_add(1, g_c, mul(1, r_s_delta_g1, neg_one))   # g_c -= &r_s_delta_g1  (-X = [r - 1] X in the prime-order subgroup)
        g_c = orc.jac_add(1, g_c, res["l"][ln])
        g_c = orc.jac_add(1, g_c, res["h"][ln])
        out.append({"a": orc.jac_to_affine(1, g_a), "b": orc.jac_to_affine(2, g2_b), "c": orc.jac_to_affine(1, g_c)})
    return out


def test_beaver_shortcut_equals_explicit_sequence(orc):
    """CPU: the algebraic shortcut the full-size GPU test feeds to the all-core checker is the explicit Beaver sequence."""
    a0, b0, c0, _, _, log_d, one = groth16_inputs(orc, 50)
    want = beaver_explicit_h(orc, a0, b0, c0, log_d, one)
    a1, b1, c1 = beaver_shortcut_lanes(orc, a0, b0, c0)
    for ln in range(4):
        assert np.array_equal(orc.witness_map_plain(a1[ln], b1[ln], c1[ln], log_d), want[ln]), ln
    # and with three parties
    a0, b0, c0, _, _, log_d, one = groth16_inputs(orc, 21, parties=3)
    want = beaver_explicit_h(orc, a0, b0, c0, log_d, one)
    a1, b1, c1 = beaver_shortcut_lanes(orc, a0, b0, c0)
    for ln in range(6):
        assert np.array_equal(orc.witness_map_plain(a1[ln], b1[ln], c1[ln], log_d), want[ln]), ln


def test_cpp_host_input_generation_matches_python_and_checker(orc):
    """CPU: tools/groth16_host.hpp builds its inputs with its own SplitMix64 streams and host field code (the role ark-ff plays
    for the reference's prover); they must equal tests/util.py / provers.py and the checker's field arithmetic."""
    from czk_amd.provers import rand_fr_canonical as prov_rand
    from test_abi import _build_host_demo
    for seed, n in ((0xC0FFEE, 40), (0xBA5E5 + 3, 257), (0xC0FFEE + 17, 1000)):
        want = rand_fr_canonical(seed, n)
        assert np.array_equal(want, prov_rand(seed, n))
        out = subprocess.run([_build_host_demo(), "inputs", str(seed), str(n)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        got = {}
        for line in out.stdout.splitlines():
            tag, *limbs = line.split()
            got.setdefault(tag, []).append([int(v, 16) for v in limbs])
        got = {k: np.array(v, dtype=np.uint64) for k, v in got.items()}
        mont = orc.fr_from_repr(want)
        nxt = np.roll(mont, -1, axis=0)
        assert np.array_equal(got["canonical"], want)
        assert np.array_equal(got["mont"], mont)
        assert np.array_equal(got["square"], orc.fr_mul(mont, mont))
        assert np.array_equal(got["sub"], orc.fr_sub(mont, nxt))
        assert np.array_equal(got["add"], orc.fr_add(mont, nxt))


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def czk():
    import czk_amd
    return czk_amd


@pytest.fixture()
def ctx(czk):
    c = czk.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_lanes_round_trip_ranges_and_errors(ctx, czk, orc):
    """czk_lanes_*: zero-filled allocation, pageable upload / download across chunk and lane boundaries, device copy, zeroing,
    out-of-range requests rejected; the lanes work as CZK_MEM_DEVICE arguments (an NTT on them equals the checker's)."""
    n = (1 << 19) + 12345                                           # 16.4 MiB per lane: more than one 16 MiB staging chunk
    L = ctx.lanes_alloc(3, n)
    assert (L.lanes, L.len) == (3, n) and L.ptr(0, 0) and L.ptr(2, n - 1) == L.ptr(0, 0) + 32 * (3 * n - 1)
    assert L.ptr(3, 0) == 0 and L.ptr(0, n) == 0
    assert not L.download(1).any()                                  # `vec![zero; n]`
    x = rand_fr_canonical(11, 2 * n + 5)
    L.upload(x, lane=0, elem=7)                                     # runs on into lanes 1 and 2
    got = np.concatenate([L.download(k) for k in range(3)])
    want = np.zeros((3 * n, 4), dtype=np.uint64)
    want[7:7 + 2 * n + 5] = x
    assert np.array_equal(got, want)
    M = ctx.lanes_alloc(1, 100)
    M.copy_from(L, 100, src_lane=1, src_elem=3)
    assert np.array_equal(M.download(), want[n + 3:n + 103])
    M.zero(10, elem=5)
    w2 = want[n + 3:n + 103].copy()
    w2[5:15] = 0
    assert np.array_equal(M.download(), w2)
    with pytest.raises(czk.CzkError):
        L.upload(x, lane=1, elem=8)                                 # 2 n + 5 elements do not fit behind (1, 8)
    with pytest.raises(czk.CzkError):
        M.download(elem=50, n=51)
    with pytest.raises(czk.CzkError):
        M.copy_from(L, 101)
    # as a device argument: coset FFT of two lanes in place
    log_d = 12
    T = ctx.lanes_alloc(2, 1 << log_d)
    v = orc.fr_from_repr(rand_fr_canonical(12, 2 << log_d)).reshape(2, 1 << log_d, 4)
    T.upload(v)
    ctx.ntt_fr(T.ptr(), log_d, czk.CZK_COSET_FFT, lanes=2, in_len=1000, mem=czk.CZK_MEM_DEVICE)
    for ln in range(2):
        assert np.array_equal(T.download(ln), orc.ntt_fr(v[ln, :1000], log_d, orc.COSET_FFT, 1000)), ln
    for h in (L, M, T):
        h.free()


def _host_demo():
    from test_abi import _build_host_demo
    return _build_host_demo()


def _read_dump(path):
    raw = np.fromfile(path, dtype=np.uint8)
    N, D, L = (int(v) for v in raw[:24].view(np.uint64))
    off = 24
    h = raw[off:off + L * D * 32].view(np.uint64).reshape(L, D, 4)
    off += L * D * 32
    pts = {}
    for name in ("h", "l", "a", "b_g1", "b_g2"):
        aw = 24 if name == "b_g2" else 12
        aff = raw[off:off + L * aw * 8].view(np.uint64).reshape(L, aw)
        off += L * aw * 8
        inf = raw[off:off + L].copy()
        off += L
        pts[name] = (aff, inf)
    proofs = []                                                    # per lane: {"a": (aff, inf), "b": ..., "c": ...} -- create_proof for public r, s
    for _ in range(L):
        pr = {}
        for key, aw in (("a", 12), ("b", 24), ("c", 12)):
            pr[key] = (raw[off:off + aw * 8].view(np.uint64).copy(), int(raw[off + aw * 8]))
            off += aw * 8 + 1
        proofs.append(pr)
    pts["proofs"] = proofs
    assert off == raw.size
    return N, D, L, h, pts


def _query_bases(ctx, czk, N, D):
    """The synthetic proving key: P_i = [k_i] G from the seeds both hosts use; b_query[0] is infinity."""
    out = {}
    for name, g, n, sd, inf_first in (("h", 1, D - 1, 1, False), ("l", 1, N, 2, False), ("a", 1, N + 1, 3, False), ("b_g1", 1, N + 1, 4, True),
                                      ("b_g2", 2, N + 1, 5, True)):
        inf = np.zeros(n, dtype=np.uint8)
        inf[0] = 1 if inf_first else 0
        out[name] = (g, ctx.fixed_base_points(g, rand_fr_canonical(0xBA5E5 + sd, n)), inf)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_constraints", [10, 1000])
def test_cpp_host_on_device_handles_matches_checker(ctx, czk, orc, tmp_path, n_constraints):
    """tools/host_demo.cpp `bench`: the complete configs[1] step (constraint evaluation, witness map with the Beaver local half
    and both opens, five MSMs on four share lanes) from a C++ process that holds only czk_lanes / czk_bases handles, against the
    checker's restatement of the same sequence on the same inputs: h bit-exact per lane, every group element equal in affine.
    (10 constraints = BASELINE configs[0]'s size.)"""
    dump = str(tmp_path / "g16.bin")
    out = subprocess.run([_host_demo(), "bench", "--constraints", str(n_constraints), "--steps", "3", "--warmup", "1", "--dump", dump],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and '"pipelined_proofs_equal": true' in out.stdout, out.stdout + out.stderr
    N, D, L, h_gpu, pts = _read_dump(dump)
    assert (N, L) == (n_constraints, 4)
    a0, b0, c0, wit, asg, log_d, one = groth16_inputs(orc, N)
    assert D == 1 << log_d
    want_h = beaver_explicit_h(orc, a0, b0, c0, log_d, one)
    for ln in range(L):
        assert np.array_equal(h_gpu[ln], want_h[ln]), ln
    scal = {"h": h_gpu, "l": wit, "a": asg, "b_g1": asg, "b_g2": asg}
    want_jac = {}
    for name, (g, bases, inf) in _query_bases(ctx, czk, N, D).items():
        want_jac[name] = []
        for ln in range(L):
            jac = orc.multi_scalar_mul(g, bases, inf, scal[name][ln].reshape(-1, 4))
            want_jac[name].append(jac)
            want, winf = orc.jac_to_affine(g, jac)
            assert bool(pts[name][1][ln]) == winf and (winf or np.array_equal(pts[name][0][ln], want)), (name, ln)
    # ... and the PROOF: every lane's share of Proof{a, b, c} (create_proof for public r, s) against the checker's restatement of prover.rs:110-178
    pk = groth16_pk_extras(lambda g, k: orc.fixed_base_points(g, k)[0])
    rs = rand_fr_canonical(0xC0FFEE + 99, 2)
    want_pf = checker_create_proof(orc, pk, want_jac, rs[0], rs[1], king_lanes=(0, 1))
    for ln in range(L):
        for key in ("a", "b", "c"):
            waff, winf = want_pf[ln][key]
            gaff, ginf = pts["proofs"][ln][key]
            assert bool(ginf) == bool(winf) and (winf or np.array_equal(gaff, waff)), ("proof", ln, key)


@pytest.mark.gpu
def test_groth16_full_size_end_to_end_matches_checker(czk, orc, tmp_path):
    """BASELINE configs[1] at its FULL size -- Groth16, SPDZ, 2 parties, 2^20 constraints, domain 2^21, four share lanes -- one
    complete step of both hosts (the Python driver bench.py times, and the torch-free C++ host on device handles) against the
    checker on the same inputs: h bit-exact on all four lanes, all 20 group elements equal in affine, and the five queries of the
    synthetic key equal to the checker's FixedBaseMSM.  The checker side is the all-core run of the C restatement
    (orc.groth16_local_par: io/oi FFTs and Pippenger of the reference) on the lanes of `beaver_shortcut_lanes` (equivalence with the
    explicit Beaver sequence: test_beaver_shortcut_equals_explicit_sequence); its answers are the committed digests of
    tests/golden/fullsize_digests.json (tests/fullsize.py; when this case is the session's live one the checker runs here, ~40 s)."""
    import torch
    import fullsize
    from czk_amd.provers import Groth16Local
    N = 1 << 20
    log_d, L = 21, 4
    D = 1 << log_d
    # the product, host 1: the Python driver
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        c2 = czk.Context(0, ts.cuda_stream)
        p = Groth16Local(czk, c2, N, 2)
        a0, _, _, _, asg, log_d2, _ = groth16_inputs(orc, N)          # the driver's lanes are the seeds' lanes
        assert log_d2 == log_d
        assert np.array_equal(p.a0.cpu().numpy().view(np.uint64), a0) and np.array_equal(p.asg.cpu().numpy().view(np.uint64), asg)
        del a0, asg
        p.step()
        torch.cuda.synchronize()
        h_py = p.ab.cpu().numpy().view(np.uint64).copy()
        res_py = {k: c2.jac_to_affine(czk.CZK_G2 if k == "b_g2" else czk.CZK_G1, v) for k, v in p.results.items()}
        assert not bool(p.chk.any().item())
        for name, (g, bases, inf) in _query_bases(c2, czk, N, D).items():   # the synthetic key both hosts register
            fullsize.expect(f"bases_groth16_{name}_2e20", {"points": bases.tobytes()}, orc)
        del p
        c2.close()
    torch.cuda.empty_cache()
    # the product, host 2: C++ on device handles
    dump = str(tmp_path / "g16_full.bin")
    out = subprocess.run([_host_demo(), "bench", "--log-n", "20", "--steps", "2", "--warmup", "1", "--dump", dump], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and '"pipelined_proofs_equal": true' in out.stdout, out.stdout + out.stderr
    _, D2, L2, h_cpp, pts = _read_dump(dump)
    assert (D2, L2) == (D, L)
    for host, h, res in (("python", h_py, res_py), ("c++", h_cpp, pts)):
        got = {f"hvec_lane{ln}": h[ln].tobytes() for ln in range(L)}
        for name in ("h", "l", "a", "b_g1", "b_g2"):
            for ln in range(L):
                assert not res[name][1][ln], (host, name, ln)
                got[f"{name}_lane{ln}"] = fullsize.affine_bytes(res[name][0][ln], res[name][1][ln])
        fullsize.expect("groth16_spdz2_2e20", got, orc)


@pytest.mark.gpu
def test_groth16_mac_msm_from_sh_equals_the_per_lane_form(czk):
    """The reference's SPDZ multi_scale_pub_group builds both of its scalar vectors from `s.sh.val` (mpc-algebra/src/share/spdz.rs:
    441-442), so the mac group share is the sh MSM again: Groth16Local(mac_msm_from_sh=True) runs one MSM per party and duplicates
    it; every group element must equal the per-lane form's (where the mac lanes carry the same values, MAC key 1)."""
    import torch
    from czk_amd.provers import Groth16Local
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        c2 = czk.Context(0, ts.cuda_stream)
        outs = []
        for flag in (False, True):
            p = Groth16Local(czk, c2, 1000, 2, mac_msm_from_sh=flag)
            p.step()
            torch.cuda.synchronize()
            r = p.expand_results(p.results)
            assert all(v.shape[0] == 4 for v in r.values())
            outs.append({k: c2.jac_to_affine(czk.CZK_G2 if k == "b_g2" else czk.CZK_G1, v) for k, v in r.items()})
            del p
        for k in outs[0]:
            assert np.array_equal(outs[0][k][0], outs[1][k][0]) and np.array_equal(outs[0][k][1], outs[1][k][1]), k
        c2.close()

"""The reference's own acceptance criterion for the whole path: the revealed proof VERIFIES (mpc-snarks/test.zsh runs groth16 / plonk / marlin and asserts
`verify_proof`, mpc-snarks/src/proof.rs:140-143).  The benchmark inputs cannot: its Groth16 key is synthetic (random points with known discrete logs), its Plonk
and Marlin inputs are work-shaped stand-ins.  Here the inputs are REAL and every check is plain big-integer Python on what the GPU path returns:

  Groth16  a key generated, as groth16/src/generator.rs:60-230 does, from toxic waste (tau, alpha, beta, gamma, delta) that the test knows (tests/groth16_real_key.py),
           handed to the prover as discrete logs (Groth16Local(key_scalars=...): the GPU builds the points).  The opened proof -- constraint evaluation, witness map
           with both opens, five MSMs on every share lane, create_proof's group steps; the parties' shares added up -- must (1) equal [a] G1, [b] G2, [c] G1 for
           the exponents the prover equations give (groth16/src/prover.rs:110-178) from the plain witness, r, s and the quotient h (points by the CPU checker's
           double-and-add, compared in affine), and (2) satisfy e(A, B) = e(alpha, beta) e(sum_i x_i gamma_abc_i, gamma) e(C, delta) (groth16/src/verifier.rs:40-62),
           which with every discrete log known is a b = alpha beta + sum_i x_i (beta u_i + alpha v_i + w_i) + c delta in Fr -- no pairing needed.  (2) holds only
           if h is the true quotient (A B - C) / Z over the reference's domain: a wrong root of unity, coset, 1/D or vanishing-polynomial factor anywhere in the witness
           map breaks it, whatever the checker's restatement says.  Also through bench.py --real-key in every layout (one process per party, split by base range).
  Plonk    a satisfied circuit (tests/polyiop_real.py), the reference's Verifier::verify (mpc-plonk/src/lib.rs:451-590): every KZG opening and the four identities.
  Marlin   a real index of a satisfied instance, the AHP verifier's decision (marlin/src/ahp/mod.rs:115-260): both sumcheck combinations, every KZG opening.

The prover sequences of provers.py / polyvm.py are shared by the checker-backed and the GPU-backed runs of the parity tests, so a slip in them is invisible there;
these tests found two (1 - s(X) with the constant on every coefficient; a and b of Marlin's third round one coefficient short)."""
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
# Plonk and Marlin: satisfied instances (tests/polyiop_real.py), the reference verifiers' equations
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_gates,parties", [(8, 3), (256, 3), (2048, 2), (1 << 18, 3)])   # the last: BASELINE configs[2]'s size
def test_plonk_proof_of_a_satisfied_circuit_verifies(n_gates, parties):
    """mpc-plonk's Prover::prove on the GPU path for a satisfied circuit, then the reference's Verifier::verify (lib.rs:511-590) on what it returns: every
    KZG opening against its commitment (with the synthetic SRS's known tau: C - [v] G == [tau - x] W, bench.verify_openings) and the four identities on the
    opened values -- public wire, gates, unit product (partial products and t(w^(k-1)) = 1), wiring."""
    import czk_amd
    from czk_amd import polyvm
    import bench
    from polyiop_real import plonk_prove_and_verify
    ctx = polyvm.shared_stream_context(czk_amd)
    B = polyvm.GpuBackend(czk_amd, ctx, parties, polyvm.plonk_max_degree(n_gates))
    plonk_prove_and_verify(polyvm, B, lambda out: bench.verify_openings(czk_amd, ctx, B, out), n_gates)
    ctx.close()


@pytest.mark.parametrize("H", [8, 64, 1024, 1 << 16, 1 << 20])   # the last: BASELINE configs[3]'s size (about half a minute; also run by bench.py)
def test_marlin_proof_of_a_satisfied_instance_verifies(H):
    """Marlin's AHP prover rounds, commitments and batched openings on the GPU path for a REAL index and a satisfied instance (the opt-in paths of
    polyvm.marlin_prove: calculate_t over the transposed matrices, the linear combinations' real coefficients), then the verifier: every KZG opening -- the
    two batched ones against the folded commitments, the degree-bound witnesses -- with the known tau (bench.verify_openings), and the AHP decision: the outer
    sumcheck combination evaluates to zero at beta and the inner one at gamma (ahp/mod.rs:168-183, 233-250), constants included."""
    import czk_amd
    from czk_amd import polyvm
    import bench
    from polyiop_real import marlin_prove_and_verify
    ctx = polyvm.shared_stream_context(czk_amd)
    lanes = 2
    B = polyvm.GpuBackend(czk_amd, ctx, lanes, polyvm.marlin_max_degree(H), lift=(1,) * lanes)    # public data on every lane: each lane is the plain prover
    marlin_prove_and_verify(polyvm, B, lambda out: bench.verify_openings(czk_amd, ctx, B, out), H)
    ctx.close()


@pytest.mark.parametrize("argv", [
    ["--log-n", "12", "--parties", "2"],                                                                              # all lanes on one GPU
    ["--gpus", "2", "--layout", "party", "--backend", "gloo", "--device", "0", "--log-n", "12", "--parties", "2"],     # one process per party, opens over torch.distributed
    ["--gpus", "3", "--layout", "party", "--backend", "gloo", "--device", "0", "--net", "czk", "--constraints", "1000", "--parties", "3"],            # ... over czk_net (shared memory)
    ["--gpus", "3", "--layout", "party", "--backend", "gloo", "--device", "0", "--net", "czk-ipc", "--scheme", "gsz", "--constraints", "1000", "--parties", "3"],   # GSZ, device mailboxes
    ["--gpus", "2", "--layout", "split", "--backend", "gloo", "--device", "0", "--constraints", "1000", "--parties", "2"],                            # one proof split by base range
])
def test_bench_real_key_proof_verifies_in_every_layout(argv):
    """bench.py --real-key: the timed proofs run under a real key and the last one is opened -- in the party layout from shares gathered over the process group,
    i.e. through the reference's own one-process-per-party layout and its network opens -- and put through the verification equation (`proof_verifies`)."""
    import json
    import os
    import subprocess
    import sys
    import util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py")] + argv + ["--real-key", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-seam-report", "--no-other-workloads"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    run = util.run_ranks if "--gpus" in argv else subprocess.run
    r = run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert r.returncode == 0, util.child_errors(r.stderr)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    pv = d["proof_verifies"]
    assert pv["proof_verifies"] and pv["proof_elements_match_prover_equations"] and pv["verification_equation_holds"] and pv["qap_identity_holds"], pv
    if "party" in argv:
        assert pv["shares_gathered_from_ranks"] == d["n_gpus"] > 1


@pytest.mark.parametrize("n_constraints,parties", [(10, 2), (1000, 3), (1 << 16, 2)])
def test_compiled_host_proof_verifies_under_a_real_key(orc, tmp_path, n_constraints, parties):
    """The torch-free, Python-free C++ host (tools/host_demo.cpp bench over tools/groth16_host.hpp: what the Rust shim's resident path does, compiled) given a
    REAL key as discrete logs (--key-file): its per-lane shares of Proof{a, b, c} for the public r, s it uses are added up and verified as above."""
    import subprocess
    import czk_amd as czk
    from test_device_handles import _host_demo, _read_dump
    N = n_constraints
    key = real_key(N, limbs_to_ints(rand_fr_canonical(0x7A11 + N, 5)))
    ks = key_scalars(key)
    kf, dump = str(tmp_path / "key.bin"), str(tmp_path / "g16.bin")
    with open(kf, "wb") as f:
        f.write(np.array([N, key["D"]], dtype=np.uint64).tobytes())
        for name in ("h", "l", "a", "b_g1", "pk_g1", "pk_g2"):
            f.write(np.ascontiguousarray(ks[name], dtype=np.uint64).tobytes())
    out = subprocess.run([_host_demo(), "bench", "--constraints", str(N), "--parties", str(parties), "--steps", "2", "--warmup", "1", "--key-file", kf, "--dump", dump],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and '"pipelined_proofs_equal": true' in out.stdout and '"mac_check_failures": 0' in out.stdout, out.stdout + out.stderr
    N2, D, L, h_lanes, pts = _read_dump(dump)
    assert (N2, D, L) == (N, key["D"], 2 * parties)
    rs = rand_fr_canonical(0xC0FFEE + 99, 2)                       # the r, s host_demo's --dump proves with
    r, s = limbs_to_ints(rs)
    ctx = czk.Context(0)
    opened = {}
    for k, g in (("a", 1), ("b", 2), ("c", 1)):
        jw = 18 if g == 1 else 36
        acc = np.zeros(jw, dtype=np.uint64)                         # the identity (z == 0)
        for j in range(parties):                                   # the parties' sh lanes: lane 2 j
            aff, inf = pts["proofs"][2 * j][k]
            acc = ctx.jac_add_mixed(g, acc, aff, bool(inf))
        opened[k] = ctx.jac_to_affine(g, acc)
    ctx.close()
    h_acc = sum(dot_mod_r(h_lanes[2 * j][:D - 1], ks["h"]) for j in range(parties)) * R_INV % R_MOD
    w0 = limbs_to_ints(rand_fr_canonical(0xC0FFEE, 1))[0]
    a_exp, b_exp, c_exp, verifies, qap = expected_exponents(key, w0, r, s, h_acc)
    for k, g, e in (("a", 1, a_exp), ("b", 2, b_exp), ("c", 1, c_exp)):
        want, winf = orc.jac_to_affine(g, orc.scalar_mul(g, orc.generator_affine(g), 0, ints_to_limbs([e], 4)[0]))
        got, ginf = opened[k]
        assert not ginf[0] and not winf and np.array_equal(np.ravel(got[0]), np.ravel(want)), (k, N, parties)
    assert verifies and qap


def test_verifiers_reject_unsatisfied_instances():
    """negative controls: the same verifier code on inputs that are NOT satisfied -- the benchmark's random Plonk polynomials; a Marlin instance whose z_b
    has one wrong entry -- must fail (the identities are checks, not tautologies of the helpers)."""
    import czk_amd
    from czk_amd import polyvm
    import bench
    import polyiop_real
    ctx = polyvm.shared_stream_context(czk_amd)
    B = polyvm.GpuBackend(czk_amd, ctx, 2, polyvm.plonk_max_degree(16))
    out = polyvm.plonk_prove(B, polyvm.plonk_inputs(B, 16))
    assert bench.verify_openings(czk_amd, ctx, B, out)["results_checked"]           # the openings are honest openings of the committed polynomials ...
    with pytest.raises(AssertionError):                                             # ... of a circuit that is not satisfied
        polyiop_real.plonk_verify(polyvm, out, [0, polyvm.unmont(out["pub_p_open"]["value"][0])], B.root_of_unity(48), 16)
    ctx.close()
    ctx = polyvm.shared_stream_context(czk_amd)
    B = polyvm.GpuBackend(czk_amd, ctx, 2, polyvm.marlin_max_degree(16), lift=(1, 1))
    inp = polyiop_real.marlin_real_inputs(B, polyvm, 16, 0x3A21)
    zb = B.download(inp["z_b"]).copy()
    zb[:, 3, 0] ^= 1                                                                # B z is wrong in one row
    inp["z_b"] = B.upload(zb)
    out = polyvm.marlin_prove(B, inp)
    with pytest.raises(AssertionError, match="outer sumcheck"):
        polyiop_real.marlin_verify(polyvm, out)
    ctx.close()

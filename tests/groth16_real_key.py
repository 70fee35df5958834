"""A REAL Groth16 key for the benchmark's squaring circuit, as discrete logs, and the verifier's equation in the exponent -- plain big-integer Python
(test infrastructure: tests/test_verify.py and bench.py's `proof_verifies` leg).  The generator, the verifier and the pairing are outside the hot path
(SURVEY.md section 8); with the toxic waste known none of them is needed to decide whether a proof verifies:

  key      groth16/src/generator.rs:60-230 with R1CStoQAP::instance_map_with_evaluation (groth16/src/r1cs_to_qap.rs:51-92): u_i(tau), v_i(tau), w_i(tau) from
           the Lagrange coefficients of the reference's domain, l_i = (beta u_i + alpha v_i + w_i) / delta, gamma_abc_i = (...) / gamma,
           h_i = tau^i Z(tau) / delta -- handed to the prover as scalars (Groth16Local(key_scalars=...): the GPU builds the points [k] G).
  prover   groth16/src/prover.rs:110-178: A = alpha + sum z_i u_i + r delta, B = beta + sum z_i v_i + s delta,
           C = s A + r B - r s delta + sum_wit z_i l_i + sum h_i (tau^i Z / delta)
  verifier groth16/src/verifier.rs:40-62: e(A, B) = e(alpha, beta) e(sum_pub x_i gamma_abc_i, gamma) e(C, delta)
           <=>  a b = alpha beta + (sum_pub x_i gamma_abc_i) gamma + c delta  (mod r)

The domain's generator comes from the reference's own constants (tests/golden/reference_constants.json: LARGE_SUBGROUP_ROOT_OF_UNITY) by the reference's rule
(ff/src/fields/mod.rs:360-367: omega = LARGE^3, squared 47 - log2 D times) -- not from the library under test and not from the checker."""
import json
import os

import numpy as np

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
R_INV = pow(1 << 256, -1, R_MOD)
_HERE = os.path.dirname(os.path.abspath(__file__))


def limbs(vals):
    """canonical integers -> (n, 4) uint64 limbs (little-endian), without a Python loop per limb"""
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def omega_for(log_d: int) -> int:
    fr = json.load(open(os.path.join(_HERE, "golden", "reference_constants.json")))["fr"]
    large = sum(int(v) << (64 * i) for i, v in enumerate(fr["LARGE_SUBGROUP_ROOT_OF_UNITY"])) * R_INV % R_MOD
    assert fr["TWO_ADICITY"] == 47 and fr["SMALL_SUBGROUP_BASE"] == 3 and fr["SMALL_SUBGROUP_BASE_ADICITY"] == 1 and log_d <= 47
    w = pow(large, 3, R_MOD)                       # omega = large_subgroup_root_of_unity ^ (small_subgroup_base ^ adicity)
    for _ in range(47 - log_d):
        w = w * w % R_MOD
    return w


def batch_inverse(v):
    """Montgomery's trick: 3 multiplications per element and one inversion"""
    pre, acc = [], 1
    for x in v:
        pre.append(acc)
        acc = acc * x % R_MOD
    inv = pow(acc, -1, R_MOD)
    out = [0] * len(v)
    for i in range(len(v) - 1, -1, -1):
        out[i] = inv * pre[i] % R_MOD
        inv = inv * v[i] % R_MOD
    return out


def real_key(N: int, toxic):
    """Discrete logs of a Groth16 key of the squaring circuit (mpc-snarks/src/proof.rs:304-344: a_i = b_i = w_i, c_i = w_{i+1}, c_{N-1} = out; variables
    [1, out | w_0 .. w_{N-1}]) for toxic waste (tau, alpha, beta, gamma, delta)."""
    log_d = (N + 2 - 1).bit_length()
    D = 1 << log_d
    tau, alpha, beta, gamma, delta = (int(v) % (R_MOD - 2) + 2 for v in toxic)
    omega = omega_for(log_d)
    assert pow(omega, D, R_MOD) == 1 and (D == 1 or pow(omega, D // 2, R_MOD) == R_MOD - 1)
    zt = (pow(tau, D, R_MOD) - 1) % R_MOD                                           # evaluate_vanishing_polynomial
    assert zt, "tau lies in the domain"
    # evaluate_all_lagrange_coefficients: L_j(tau) = Z(tau) omega^j / (D (tau - omega^j))
    wp, wj = [], 1
    for _ in range(N + 2):
        wp.append(wj)
        wj = wj * omega % R_MOD
    k0 = zt * pow(D, -1, R_MOD) % R_MOD
    lag = [k0 * wj % R_MOD * iv % R_MOD for wj, iv in zip(wp, batch_inverse([(tau - wj) % R_MOD for wj in wp]))]
    nv = N + 2
    a, b, c = [0] * nv, [0] * nv, [0] * nv
    a[0], a[1] = lag[N], lag[N + 1]                                                  # the instance copy rows (r1cs_to_qap.rs:75-80)
    a[2:] = lag[:N]
    b[2:] = lag[:N]
    c[3:] = lag[:N - 1]
    c[1] = lag[N - 1]
    g_inv, dl_inv = pow(gamma, -1, R_MOD), pow(delta, -1, R_MOD)
    lin = [(beta * x + alpha * y + w) % R_MOD for x, y, w in zip(a, b, c)]
    h, t = [], zt * dl_inv % R_MOD
    for _ in range(D - 1):
        h.append(t)
        t = t * tau % R_MOD
    return {"tau": tau, "alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta, "D": D, "log_d": log_d, "zt": zt, "omega": omega,
            "a": a, "b": b, "c": c, "l": [v * dl_inv % R_MOD for v in lin], "gamma_abc": [lin[i] * g_inv % R_MOD for i in range(2)], "h": h}


def key_scalars(key):
    """what Groth16Local(key_scalars=...) takes: the queries from index 1 on, [alpha, beta, delta, a_query[0]] in G1, [beta, delta] in G2"""
    assert key["b"][0] == 0 and key["b"][1] == 0       # b_query[0], b_query[1] are infinity in a real key: what the prover's infinity flags assume
    return {"h": limbs(key["h"]), "l": limbs(key["l"][2:]), "a": limbs(key["a"][1:]), "b_g1": limbs(key["b"][1:]), "b_g2": limbs(key["b"][1:]),
            "pk_g1": limbs([key["alpha"], key["beta"], key["delta"], key["a"][0]]), "pk_g2": limbs([key["beta"], key["delta"]])}


def expected_exponents(key, w0: int, r: int, s: int, h_acc: int):
    """(a, b, c, verifies, qap_holds) for the circuit's plain witness w_i = w0^(2^i), public r, s and h_acc = sum_i h_i (tau^i Z(tau) / delta) -- the h MSM in
    the exponent, from the quotient the prover computed."""
    N = len(key["a"]) - 2
    w = [w0 % R_MOD]
    for _ in range(N):
        w.append(w[-1] * w[-1] % R_MOD)
    z = [1, w[N]] + w[:N]
    al, be, de = key["alpha"], key["beta"], key["delta"]
    dot = lambda q: sum(zi * qi for zi, qi in zip(z, q)) % R_MOD   # noqa: E731
    za, zb, zc = dot(key["a"]), dot(key["b"]), dot(key["c"])
    a_exp = (al + za + r * de) % R_MOD                                                          # prover.rs:124-137
    b_exp = (be + zb + s * de) % R_MOD                                                          # :141-160
    l_acc = sum(zi * qi for zi, qi in zip(z[2:], key["l"][2:])) % R_MOD
    c_exp = (s * a_exp + r * b_exp - r * s % R_MOD * de + l_acc + h_acc) % R_MOD                # :162-169
    pub = (key["gamma_abc"][0] + w[N] * key["gamma_abc"][1]) % R_MOD                           # prepare_inputs (verifier.rs:22-37), x = [1, out]
    verifies = a_exp * b_exp % R_MOD == (al * be + pub * key["gamma"] + c_exp * de) % R_MOD
    qap = (za * zb - zc) % R_MOD == h_acc * de % R_MOD                                          # A(tau) B(tau) - C(tau) = h(tau) Z(tau)
    return a_exp, b_exp, c_exp, verifies, qap

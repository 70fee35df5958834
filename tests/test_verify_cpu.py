"""CPU half of tests/test_verify.py: the real-key generator and the verification equation in the exponent (tests/groth16_real_key.py) hold together on the
CHECKER's witness map -- so the GPU tests that use them test the GPU path and not the test."""
import pytest

from groth16_real_key import R_INV, R_MOD, expected_exponents, key_scalars, omega_for, real_key
from util import ints_to_limbs, limbs_to_ints, rand_fr_canonical


@pytest.mark.parametrize("N", [2, 10, 333, 1022])
def test_real_key_and_verification_equation_on_the_checkers_witness_map(orc, N):
    key = real_key(N, limbs_to_ints(rand_fr_canonical(0x7A11 + N, 5)))
    D, ld = key["D"], key["log_d"]
    assert key["omega"] == limbs_to_ints(orc.domain_constants(ld)["group_gen"].reshape(1, 4))[0] * R_INV % R_MOD == omega_for(ld)
    ks = key_scalars(key)
    assert ks["h"].shape == (D - 1, 4) and ks["l"].shape == (N, 4) and ks["a"].shape == (N + 1, 4) and ks["pk_g1"].shape == (4, 4)
    w = [limbs_to_ints(rand_fr_canonical(0xC0FFEE, 1))[0]]
    for _ in range(N):
        w.append(w[-1] * w[-1] % R_MOD)
    a = w[:N] + [1, w[N]] + [0] * (D - N - 2)
    b = w[:N] + [0] * (D - N)
    c = w[1:N + 1] + [0] * (D - N)
    tom = lambda v: orc.fr_from_repr(ints_to_limbs(v, 4))   # noqa: E731
    h = [v * R_INV % R_MOD for v in limbs_to_ints(orc.witness_map_plain(tom(a), tom(b), tom(c), ld))]
    assert h[D - 1] == 0
    h_acc = sum(hi * qi for hi, qi in zip(h, key["h"])) % R_MOD
    r, s = limbs_to_ints(rand_fr_canonical(0xC0FFEE + 77 + N, 2))
    _, _, _, verifies, qap = expected_exponents(key, w[0], r, s, h_acc)
    assert verifies and qap
    # and a wrong quotient does not verify: the equation is a check, not an identity of the helper
    _, _, _, verifies_bad, qap_bad = expected_exponents(key, w[0], r, s, (h_acc + 1) % R_MOD)
    assert not verifies_bad and not qap_bad

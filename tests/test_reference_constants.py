"""The constants this repository transcribed by hand (oracle/pyref.py, oracle/czk_oracle.c, csrc/field.h, fqu.h, msm.hip,
ntt.hip) equal the reference's parameter files.  tests/golden/reference_constants.json is DATA parsed out of
curves/bls12_377/src/{fields/fr.rs, fields/fq.rs, fields/fq2.rs, curves/g1.rs, curves/g2.rs} by
tools/check_constants_vs_reference.py in the build container; where /root/reference is present the parse is repeated."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_constants", os.path.join(ROOT, "tools", "check_constants_vs_reference.py"))
chk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(chk)


def test_repository_constants_equal_the_reference_parameter_files():
    fx = json.load(open(chk.FIXTURE))
    bad, n = chk.compare(fx)
    assert n >= 50 and not bad, bad


def test_fixture_is_what_the_reference_text_says():
    import pytest
    if not os.path.isdir("/root/reference/curves/bls12_377/src"):
        pytest.skip("reference tree not present on this machine")
    now = chk.read_reference("/root/reference")
    assert json.loads(json.dumps(now)) == json.load(open(chk.FIXTURE))


def test_reference_constants_are_self_consistent():
    """The KATs the reference itself relies on (fields/tests.rs:352-395), evaluated on the parsed data."""
    fx = json.load(open(chk.FIXTURE))
    e = chk.expected_from_fixture(fx)
    for f, bits in (("fr", 256), ("fq", 384)):
        p = e[f + ".MODULUS"]
        assert e[f + ".R"] == (1 << bits) % p and e[f + ".R2"] == (1 << (2 * bits)) % p
        assert (p * e[f + ".INV"]) % (1 << 64) == (1 << 64) - 1
        s, t = e[f + ".TWO_ADICITY"], e[f + ".T"]
        assert p - 1 == t << s and t & 1
        rinv = pow(1 << bits, -1, p)
        g = e[f + ".GENERATOR"] * rinv % p
        assert pow(g, t, p) == e[f + ".TWO_ADIC_ROOT_OF_UNITY"] * rinv % p
    assert e["fr.GENERATOR"] * pow(1 << 256, -1, e["fr.MODULUS"]) % e["fr.MODULUS"] == 22   # the comment in fr.rs says 11; the limbs say 22
    q = e["fq.MODULUS"]
    x, y = e["g1.G1_GENERATOR_X"], e["g1.G1_GENERATOR_Y"]
    assert (y * y - x * x * x - 1) % q == 0

"""Call-site audit of the Plonk / Marlin drivers: collaborative-zksnark_amd/polyvm.py, run on a shape-only counting backend, must emit
exactly the operations of tests/golden/plonk_marlin_callsites.json -- a list written by hand from the reference's source
(mpc-plonk/src/lib.rs:110-448, marlin/src/ahp/prover.rs:300-704, marlin/src/lib.rs:170-318, marlin/src/ahp/mod.rs:115-260,
poly-commit/src/marlin/marlin_pc/mod.rs:245-330), in order, with the sizes and share / public lane kinds the reference has."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _pow2(n):
    return 1 << max(0, (int(n) - 1).bit_length())


def _expected(spec, env, lanes):
    out = []
    for site in spec["sites"]:
        if "fused" in site or "unmodelled" in site:
            continue
        for op in site["ops"]:
            if op[0] == "transcript_point":
                out.append(("transcript_point",))
                continue
            size = int(eval(op[1], {"__builtins__": {}, "pow2": _pow2}, dict(env)))
            ln = lanes if op[-1] == "s" else 1
            out.append((op[0], size, ln) + ((op[2],) if op[0] == "ntt" else ()))
    return out


def _emitted(log):
    out = []
    for e in log:
        if e[0] == "ntt":
            out.append(("ntt", e[1], e[2], e[3]))
        elif e[0] == "transcript_point":
            out.append(e)
        else:
            out.append(tuple(e[:3]))
    return out


@pytest.mark.parametrize("prover,size", [("plonk", 64), ("plonk", 16), ("marlin", 100), ("marlin", 31)])
def test_drivers_emit_the_reference_call_sites(prover, size):
    import czk_amd  # noqa: F401
    from czk_amd import polyvm
    from shape_backend import ShapeBackend
    spec = json.load(open(os.path.join(HERE, "golden", "plonk_marlin_callsites.json")))[prover]
    if prover == "plonk":
        lanes, env = 3, {"G": size, "W": 3 * size}
        B = ShapeBackend(lanes)
        inp = polyvm.plonk_inputs(B, size)
        B.log = []
        polyvm.plonk_prove(B, inp)
    else:
        H = _pow2(size)
        lanes, env = 4, {"H": H, "K": H, "X": 2, "zk": 1}
        B = ShapeBackend(lanes, lift=(1, 1, 0, 0))
        inp = polyvm.marlin_inputs(B, size)
        B.log = []
        polyvm.marlin_prove(B, inp)
    want, got = _expected(spec, env, lanes), _emitted(B.log)
    for i, (w, g) in enumerate(zip(want, got)):
        assert w == g, f"{prover}: operation #{i}: reference {w}, driver {g}"
    assert len(want) == len(got), (len(want), len(got), want[len(got):], got[len(want):])


def test_callsite_file_cites_existing_reference_lines():
    """Where the reference tree is present (the build container), every cited file exists and has at least the cited lines."""
    import re
    root = "/root/reference"
    if not os.path.isdir(root):
        pytest.skip("reference tree not present on this box")
    spec = json.load(open(os.path.join(HERE, "golden", "plonk_marlin_callsites.json")))
    n = 0
    for prover in ("plonk", "marlin"):
        for site in spec[prover]["sites"]:
            for path, line in re.findall(r"([\w\-/\.]+\.rs):(\d+)", site["ref"]):
                full = os.path.join(root, path)
                assert os.path.exists(full), path
                assert sum(1 for _ in open(full)) >= int(line), (path, line)
                n += 1
    assert n > 80

#!/usr/bin/env python3
"""Pins every hand-transcribed BLS12-377 constant of this repository to the TEXT of the reference's parameter files.

Build-container tool (needs /root/reference; it is absent on the GPU box).  It parses
    curves/bls12_377/src/fields/{fr,fq,fq2}.rs   and   curves/bls12_377/src/curves/{g1,g2}.rs
(`const NAME: BigInteger = BigInteger([..])`, `Some(BigInteger([..]))`, `const NAME: u32|u64 = ..`, `field_new!(Fq, "..")`),
writes what it found to tests/golden/reference_constants.json (DATA only: names and numbers, plus the SHA-256 of each parsed
file), and then asserts that the constants carried by
    oracle/pyref.py, oracle/czk_oracle.c            (the checker)
    collaborative-zksnark_amd/csrc/field.h, fqu.h, msm.hip, ntt.hip   (the product)
equal them.  tests/test_reference_constants.py repeats the second half against the committed fixture on any machine and, where
/root/reference exists, the first half too.

    python tools/check_constants_vs_reference.py [--reference /root/reference] [--write]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILES = {"fr": "curves/bls12_377/src/fields/fr.rs", "fq": "curves/bls12_377/src/fields/fq.rs", "fq2": "curves/bls12_377/src/fields/fq2.rs",
             "g1": "curves/bls12_377/src/curves/g1.rs", "g2": "curves/bls12_377/src/curves/g2.rs"}
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_constants.json")


def _num(tok: str) -> int:
    tok = tok.strip().replace("_", "")
    tok = re.sub(r"u(32|64)$", "", tok)
    return int(tok, 16) if tok.lower().startswith("0x") else int(tok)


def parse_rust_constants(text: str) -> dict:
    """{NAME: int | [limbs] | "decimal string"} for the constant forms used by the five parameter files."""
    text = re.sub(r"//[^\n]*", "", text)   # comments hold decimal renderings that are not authoritative (fr.rs says GENERATOR = 11)
    out = {}
    for m in re.finditer(r"const\s+(\w+)\s*:\s*[^=]*?=\s*(?:Some\()?\s*BigInteger\(\[(.*?)\]\)", text, flags=re.S):
        out[m.group(1)] = [_num(t) for t in m.group(2).split(",") if t.strip()]
    for m in re.finditer(r"const\s+(\w+)\s*:\s*(?:u32|u64)\s*=\s*([0-9a-fA-Fx_]+(?:u32|u64)?)\s*;", text):
        out[m.group(1)] = _num(m.group(2))
    for m in re.finditer(r"const\s+(\w+)\s*:\s*Option<u32>\s*=\s*Some\((\d+)\)", text):
        out[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"const\s+(\w+)\s*:\s*(?:Fq|Fr)\s*=\s*field_new!\(\s*(?:Fq|Fr)\s*,\s*\"(-?\d+)\"\s*\)", text):
        out[m.group(1)] = m.group(2)
    for m in re.finditer(r"const\s+(\w+)\s*:\s*&'static\s*\[u64\]\s*=\s*&\[(.*?)\]", text, flags=re.S):
        out[m.group(1)] = [_num(t) for t in m.group(2).split(",") if t.strip()]
    # g2.rs COEFF_B = field_new!(Fq2, FQ_ZERO, field_new!(Fq, "..."))
    m = re.search(r"const\s+COEFF_B\s*:\s*Fq2\s*=\s*field_new!\(\s*Fq2\s*,\s*(\w+)\s*,\s*field_new!\(\s*Fq\s*,\s*\"(\d+)\"\s*\)", text, flags=re.S)
    if m:
        out["COEFF_B"] = [m.group(1), m.group(2)]
    m = re.search(r"const\s+COEFF_B\s*:\s*Fq\s*=\s*(\w+)\s*;", text)
    if m:
        out["COEFF_B"] = m.group(1)
    return out


def read_reference(ref_root: str) -> dict:
    fx = {"_source": "alex-ozdemir/collaborative-zksnark, curves/bls12_377/src (parsed by tools/check_constants_vs_reference.py)", "_sha256": {}}
    for key, rel in REF_FILES.items():
        txt = open(os.path.join(ref_root, rel)).read()
        fx["_sha256"][rel] = hashlib.sha256(txt.encode()).hexdigest()
        fx[key] = parse_rust_constants(txt)
    return fx


def limbs_int(limbs, bits=64):
    return sum(int(v) << (bits * i) for i, v in enumerate(limbs))


# ------------------------------------------------------------------------------------------------------------------
# what this repository carries
# ------------------------------------------------------------------------------------------------------------------
def _c_arrays(path: str, pattern: str, bits: int) -> dict:
    txt = open(path).read()
    out = {}
    for m in re.finditer(pattern, txt, flags=re.S):
        vals = [int(re.sub(r"(?i)(ull|u)$", "", t.strip()), 0) for t in m.group(2).split(",") if t.strip()]
        out[m.group(1)] = limbs_int(vals, bits)
    return out


def repo_constants() -> dict:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyref
    got = {}
    L = pyref.limbs_to_int
    got["pyref"] = {
        "fr.MODULUS": L(pyref.FR_MODULUS_LIMBS), "fr.R": L(pyref.FR_R_LIMBS), "fr.R2": L(pyref.FR_R2_LIMBS), "fr.INV": pyref.FR_INV,
        "fr.GENERATOR": L(pyref.FR_GENERATOR_LIMBS), "fr.TWO_ADIC_ROOT_OF_UNITY": L(pyref.FR_TWO_ADIC_ROOT_LIMBS),
        "fr.LARGE_SUBGROUP_ROOT_OF_UNITY": L(pyref.FR_LARGE_SUBGROUP_ROOT_LIMBS), "fr.T": pyref.FR_T,
        "fq.MODULUS": L(pyref.FQ_MODULUS_LIMBS), "fq.R": L(pyref.FQ_R_LIMBS), "fq.R2": L(pyref.FQ_R2_LIMBS), "fq.INV": pyref.FQ_INV,
        "fq.GENERATOR": L(pyref.FQ_GENERATOR_LIMBS), "fq.TWO_ADIC_ROOT_OF_UNITY": L(pyref.FQ_TWO_ADIC_ROOT_LIMBS), "fq.TWO_ADICITY": pyref.FQ_TWO_ADICITY,
        "fq.T": pyref.FQ_T, "fq2.NONRESIDUE": pyref.FQ2_NONRESIDUE - pyref.Q_MOD,
        "g1.G1_GENERATOR_X": pyref.G1_GEN[0], "g1.G1_GENERATOR_Y": pyref.G1_GEN[1], "g1.COFACTOR": pyref.G1_COFACTOR,
        "g2.COEFF_B.c1": pyref.G2_B[1], "g2.COEFF_B.c0": pyref.G2_B[0],
        "g2.G2_GENERATOR_X_C0": pyref.G2_GEN[0][0], "g2.G2_GENERATOR_X_C1": pyref.G2_GEN[0][1],
        "g2.G2_GENERATOR_Y_C0": pyref.G2_GEN[1][0], "g2.G2_GENERATOR_Y_C1": pyref.G2_GEN[1][1],
    }
    c = _c_arrays(os.path.join(ROOT, "oracle", "czk_oracle.c"), r"static const uint64_t (\w+)\[\d+\]\s*=\s*\{(.*?)\};", 64)
    ctxt = open(os.path.join(ROOT, "oracle", "czk_oracle.c")).read()
    defs = {m.group(1): int(re.sub(r"(?i)ull$", "", m.group(2)), 0) for m in re.finditer(r"#define\s+(\w+)\s+(\d+ULL|\d+)\s*$", ctxt, flags=re.M)}
    got["czk_oracle.c"] = {"fr.MODULUS": c["fr_MODULUS"], "fr.R": c["fr_R"], "fr.R2": c["fr_R2"], "fr.INV": defs["fr_INV"], "fr.GENERATOR": c["fr_GENERATOR"],
                           "fr.LARGE_SUBGROUP_ROOT_OF_UNITY": c["fr_LARGE_ROOT"], "fr.TWO_ADICITY": defs["FR_TWO_ADICITY"],
                           "fq.MODULUS": c["fq_MODULUS"], "fq.R": c["fq_R"], "fq.R2": c["fq_R2"], "fq.INV": defs["fq_INV"]}
    # product: field.h (32-bit limb views inside struct FrParams / FqParams)
    ftxt = open(os.path.join(ROOT, "collaborative-zksnark_amd", "csrc", "field.h")).read()
    prod = {}
    for struct, key in (("FrParams", "fr"), ("FqParams", "fq")):
        body = ftxt[ftxt.index("struct " + struct):]
        body = body[:body.index("\n};") + 3]
        for fn, name in (("p", "MODULUS"), ("r", "R"), ("r2", "R2")):
            m = re.search(r"u32 " + fn + r"\(int i\)\s*\{\s*constexpr u32 m\[\d+\]\s*=\s*\{(.*?)\};", body, flags=re.S)
            prod[f"{key}.{name}"] = limbs_int([int(t.strip().rstrip("u"), 16) for t in m.group(1).split(",") if t.strip()], 32)
    got["field.h"] = prod
    # product: fqu.h 28-bit limbs of the Fq modulus; msm.hip generators (Montgomery, 32-bit limbs); ntt.hip LARGE root
    utxt = open(os.path.join(ROOT, "collaborative-zksnark_amd", "csrc", "fqu.h")).read()
    m = re.search(r"u32 fqu_p\(int i\)\s*\{\s*constexpr u32 m\[14\]\s*=\s*\{(.*?)\};", utxt, flags=re.S)
    got["fqu.h"] = {"fq.MODULUS": limbs_int([int(t.strip().rstrip("u"), 16) for t in m.group(1).split(",") if t.strip()], 28)}
    mtxt = open(os.path.join(ROOT, "collaborative-zksnark_amd", "csrc", "msm.hip")).read()
    arr = {m.group(1): limbs_int([int(t.strip().rstrip("u"), 16) for t in m.group(2).split(",") if t.strip()], 32)
           for m in re.finditer(r"const u32 (\w+)\[12\]\s*=\s*\{(.*?)\};", mtxt, flags=re.S)}
    got["msm.hip (Montgomery form)"] = {"g1.G1_GENERATOR_X": arr["gx"], "g1.G1_GENERATOR_Y": arr["gy"], "g2.G2_GENERATOR_X_C0": arr["x0"],
                                        "g2.G2_GENERATOR_X_C1": arr["x1"], "g2.G2_GENERATOR_Y_C0": arr["y0"], "g2.G2_GENERATOR_Y_C1": arr["y1"]}
    ntxt = open(os.path.join(ROOT, "collaborative-zksnark_amd", "csrc", "ntt.hip")).read()
    m = re.search(r"const u32 lr\[8\]\s*=\s*\{(.*?)\};", ntxt, flags=re.S)
    got["ntt.hip"] = {"fr.LARGE_SUBGROUP_ROOT_OF_UNITY": limbs_int([int(t.strip().rstrip("u"), 16) for t in m.group(1).split(",") if t.strip()], 32)}
    return got


def expected_from_fixture(fx: dict) -> dict:
    """name -> integer, from the parsed reference data."""
    e = {}
    for key in ("fr", "fq"):
        for name, v in fx[key].items():
            e[f"{key}.{name}"] = limbs_int(v) if isinstance(v, list) else int(v)
    e["fq2.NONRESIDUE"] = int(fx["fq2"]["NONRESIDUE"])
    for name in ("G1_GENERATOR_X", "G1_GENERATOR_Y"):
        e["g1." + name] = int(fx["g1"][name])
    e["g1.COFACTOR"] = limbs_int(fx["g1"]["COFACTOR"])   # little-endian u64 words: [0x0, 0x170b5d4430000000]
    assert fx["g1"]["COEFF_B"] == "FQ_ONE" and fx["g2"]["COEFF_B"][0] == "FQ_ZERO"
    e["g2.COEFF_B.c0"], e["g2.COEFF_B.c1"] = 0, int(fx["g2"]["COEFF_B"][1])
    for name in ("G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1"):
        e["g2." + name] = int(fx["g2"][name])
    return e


def compare(fx: dict) -> list[str]:
    """Returns the list of mismatches between the repository's constants and the reference data (empty = pinned)."""
    exp = expected_from_fixture(fx)
    q = exp["fq.MODULUS"]
    bad = []
    n = 0
    for where, consts in repo_constants().items():
        for name, val in consts.items():
            want = exp[name]
            if "Montgomery" in where:
                want = want * (1 << 384) % q
            n += 1
            if val != want:
                bad.append(f"{where}: {name} = {val:#x}, reference says {want:#x}")
    return bad, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--write", action="store_true", help="(re)write tests/golden/reference_constants.json")
    args = ap.parse_args()
    fx = read_reference(args.reference)
    if args.write:
        os.makedirs(os.path.dirname(FIXTURE), exist_ok=True)
        json.dump(fx, open(FIXTURE, "w"), indent=1, sort_keys=True)
        print("wrote", FIXTURE)
    bad, n = compare(fx)
    for b in bad:
        print("MISMATCH", b)
    print(f"{n - len(bad)}/{n} constants of oracle/ and csrc/ equal the reference's parameter files")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

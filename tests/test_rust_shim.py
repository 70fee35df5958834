"""The Rust side of the drop-in boundary (rust/czk-sys, rust/czk).  No Rust toolchain exists in this image, so the checks are
structural: the raw bindings are GENERATED from include/czk.h and must be current, every C entry point has exactly one
declaration, the safe crate only calls functions that exist, and the sources are at least lexically well-formed."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_czk_sys_is_generated_from_the_header_and_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "rust/czk-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py"


def test_every_header_symbol_is_bound_exactly_once():
    sys.path.insert(0, ROOT)
    import czk_amd
    src = open(os.path.join(ROOT, "rust", "czk-sys", "src", "lib.rs")).read()
    for sym in czk_amd.header_symbols():
        assert len(re.findall(r"pub fn %s\(" % sym, src)) == 1, sym


def _strip(src):
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(\\.|[^"\\])*"', '""', src)
    return re.sub(r"'(\\.|[^'\\])'", "' '", src)


def test_rust_sources_are_lexically_balanced_and_call_existing_functions():
    sysfns = set(re.findall(r"pub fn (czk_\w+)\(", open(os.path.join(ROOT, "rust", "czk-sys", "src", "lib.rs")).read()))
    consts = set(re.findall(r"pub const (CZK_\w+)", open(os.path.join(ROOT, "rust", "czk-sys", "src", "lib.rs")).read()))
    for rel in ("rust/czk/src/lib.rs", "rust/czk-sys/src/lib.rs", "rust/czk-sys/build.rs"):
        src = _strip(open(os.path.join(ROOT, rel)).read())
        for a, b in ("{}", "()", "[]"):
            assert src.count(a) == src.count(b), (rel, a)
    used = set(re.findall(r"sys::(czk_\w+)\(", open(os.path.join(ROOT, "rust", "czk", "src", "lib.rs")).read()))
    assert used and used <= sysfns, used - sysfns
    used_c = set(re.findall(r"sys::(CZK_\w+)", open(os.path.join(ROOT, "rust", "czk", "src", "lib.rs")).read()))
    assert used_c <= consts, used_c - consts

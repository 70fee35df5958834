#!/usr/bin/env python3
"""Generates rust/czk-sys/src/lib.rs -- the raw `extern "C"` block -- from include/czk.h, one declaration per C declaration, so
the two cannot drift (tests/test_rust_shim.py re-runs the generator and compares).

    python tools/gen_rust_sys.py [--check]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "czk.h")
OUT = os.path.join(ROOT, "rust", "czk-sys", "src", "lib.rs")

TYPES = {
    "int": "c_int", "unsigned": "c_uint", "size_t": "usize", "void": "()", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8",
    "char": "c_char", "double": "f64", "long": "c_long",
}


def opaque_types(text: str):
    """`typedef struct czk_x czk_x;` declarations of the header, in order: opaque handles."""
    return re.findall(r"typedef struct (czk_\w+) \1;", text)


def rust_type(c: str) -> str:
    """C declarator type -> Rust.  Each `*` takes the constness of what stands to its LEFT: `const T*` -> *const T, `T* const*` ->
    *const *mut T, `const T* const*` -> *const *const T (a `const` right of the last star qualifies the parameter itself: dropped)."""
    c = c.strip()
    parts = [p.strip() for p in c.split("*")]
    base = parts[0].split()
    const = "const" in base
    t = TYPES[" ".join(b for b in base if b != "const")]
    if len(parts) == 1:
        return t
    if t == "()":
        t = "c_void"
    out = t
    for level in range(1, len(parts)):
        out = ("*const " if const else "*mut ") + out
        const = "const" in parts[level].split()
    return out


def parse_header(text: str):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    enums = []
    for m in re.finditer(r"typedef enum\s*\{(.*?)\}\s*(\w+);", text, flags=re.S):
        for item in m.group(1).split(","):
            item = item.strip()
            if item:
                name, val = [v.strip() for v in item.split("=")]
                enums.append((m.group(2), name, int(val)))
    funcs = []
    for m in re.finditer(r"^([\w\s\*]+?)\b(czk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(2), mm.group(1).strip()))
        funcs.append((name, ret, params))
    return enums, funcs


def generate() -> str:
    text = open(HEADER).read()
    enums, funcs = parse_header(text)
    opaques = opaque_types(text)
    for o in opaques:
        TYPES[o] = o
    out = ["//! Raw bindings of `libczk_hip.so` -- GENERATED from include/czk.h by tools/gen_rust_sys.py; do not edit.",
           "//! One `extern \"C\"` declaration per C declaration; constants mirror the C enums.  Safe wrappers live in the `czk` crate.",
           "#![allow(non_camel_case_types)]", "use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};", ""]
    for opaque in opaques:
        out += ["#[repr(C)]", f"pub struct {opaque} {{", "    _private: [u8; 0],", "}"]
    out.append("")
    for ty, name, val in enums:
        out.append(f"pub const {name}: c_int = {val}; // {ty}")
    for name, val in re.findall(r"^#define (CZK_[A-Z0-9_]+) (\d+)\s*$", text, flags=re.M):
        out.append(f"pub const {name}: c_int = {val}; // #define")
    out += ["", "#[link(name = \"czk_hip\")]", "extern \"C\" {"]
    for name, ret, params in funcs:
        ps = ", ".join(f"{('type_' if p == 'type' else p)}: {rust_type(t)}" for p, t in params)
        r = rust_type(ret)
        out.append(f"    pub fn {name}({ps})" + ("" if r == "()" else f" -> {r}") + ";")
    out.append("}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    src = generate()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == src else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(src)
    print("wrote", OUT, f"({src.count('pub fn ')} functions)")

"""CPU-only checks of the drop-in boundary: libczk_hip.so loads without a GPU and exports every symbol that
include/czk.h declares; calls that need a GPU fail loudly (no CPU fallback)."""
import ctypes as C

import pytest


def test_library_exports_every_declared_symbol():
    import czk_amd
    hs, es = czk_amd.header_symbols(), czk_amd.exported_symbols()
    assert len(hs) >= 20
    assert hs == es, sorted(set(hs) - set(es))
    assert b"gfx950" in czk_amd.lib().czk_version()


def test_product_never_touches_the_oracle():
    # the product package must not import, link or reference anything under oracle/
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "collaborative-zksnark_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "czk_oracle" not in txt and "import orc" not in txt and "pyref" not in txt, fn
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libczk_hip.so")], capture_output=True, text=True).stdout
    assert "czk_oracle" not in out


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import czk_amd
    with pytest.raises(czk_amd.CzkError):
        czk_amd.Context(0)
    # null-context calls are rejected, not crashed
    assert czk_amd.lib().czk_ntt_fr(None, None, C.c_uint(3), C.c_size_t(1), 0, C.c_size_t(8), 0) == 3


def _build_host_demo(lab: bool = False):
    """tools/host_demo.bin, (re)compiled with -Wall when it is older than its sources or the library.  lab=True: tools/host_demo_lab.bin, the same
    sources linked against libczk_hip_lab.so (the lab build reads the CZK_* environment switches: tests/test_chaos.py)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "tools", "host_demo_lab.bin" if lab else "host_demo.bin")
    pkg = os.path.join(root, "collaborative-zksnark_amd")
    lib = "czk_hip_lab" if lab else "czk_hip"
    deps = [os.path.join(root, "tools", f) for f in ("host_demo.cpp", "groth16_host.hpp", "polyvm_host.hpp")] + \
           [os.path.join(root, "include", f) for f in ("czk.h", "czk.hpp")] + [os.path.join(pkg, f"lib{lib}.so")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "host_demo.cpp"),
                               "-I" + os.path.join(root, "tools"), "-L" + pkg, "-l" + lib, "-lpthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib",
                               "-Wl,-rpath-link,/opt/rocm/lib", "-o", out])
    return out


def test_cpp_host_mirror_compiles_against_the_abi():
    # include/czk.hpp (the C++ mirror of the reference's EvaluationDomain / VariableBaseMSM / MpcField surface)
    # must compile with plain g++ against czk.h and link against libczk_hip.so
    import os
    exe = _build_host_demo()
    os.remove(exe)                      # this test compiles; the others reuse the binary
    assert os.path.exists(_build_host_demo())


@pytest.mark.gpu
def test_cpp_host_mirror_runs_on_gpu():
    import subprocess
    out = subprocess.run([_build_host_demo()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host_demo OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_share_lift_matches_checker_for_king_and_other_party(orc):
    """MpcField vectors mixing Public and Shared entries through the C++ mirror's EvaluationDomain (include/czk.hpp), as the king
    and as a non-king party, against the checker: per SURVEY a18 the result must equal the reference transform of the lanes in
    which every Public(x) was lifted to (king ? x : 0, mac_share * x), mac_share = 1 on the king and 0 elsewhere
    (mpc-algebra/src/share/spdz.rs:30-37, 204-208)."""
    import subprocess
    import numpy as np
    out = subprocess.run([_build_host_demo(), "dump-lift"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = np.array([(0x9e3779b97f4a7c15 * (i + 1) % (1 << 64)) >> 7 for i in range(26)], dtype=np.uint64)
    x = orc.fr_from_repr(np.stack([raw, np.zeros(26, np.uint64), np.zeros(26, np.uint64), np.zeros(26, np.uint64)], axis=1))
    cases = out.stdout.split("case ")[1:]
    assert len(cases) == 4
    for block in cases:
        head, *lines = block.strip().splitlines()
        king = "king=1" in head
        kind = orc.COSET_FFT if "coset_fft" in head else orc.IFFT
        got = {"sh": [], "mac": []}
        for ln in lines:
            tag, *limbs = ln.split()
            got[tag].append([int(v, 16) for v in limbs])
        sh_in, mac_in = np.zeros((13, 4), np.uint64), np.zeros((13, 4), np.uint64)
        for i in range(13):
            if i % 3 != 1:
                sh_in[i], mac_in[i] = x[i], x[13 + i]
            elif king:
                sh_in[i], mac_in[i] = x[i], x[i]
        assert np.array_equal(np.array(got["sh"], dtype=np.uint64), orc.ntt_fr(sh_in, 4, kind, 13)), head
        assert np.array_equal(np.array(got["mac"], dtype=np.uint64), orc.ntt_fr(mac_in, 4, kind, 13)), head


def test_host_side_field_code_matches_checker(orc):
    """czk_jac_to_affine is host arithmetic from the same field.h the kernels use (Montgomery multiply, dedicated
    squaring, Fermat inverse, Fq2): check it against the checker's From<Projective> on random Jacobian points."""
    import numpy as np
    import czk_amd
    import pyref as P
    import random
    rng = random.Random(5)
    for g, F, gen, width in ((1, P.F1, P.G1_GEN, 6), (2, P.F2, P.G2_GEN, 12)):
        jacs = []
        for _ in range(6):
            pt = P.ec_mul(F, rng.randrange(1, P.R_MOD), gen)
            z = rng.randrange(1, P.Q_MOD) if g == 1 else (rng.randrange(1, P.Q_MOD), rng.randrange(P.Q_MOD))
            z2 = F.mul(z, z)
            x, y = F.mul(pt[0], z2), F.mul(pt[1], F.mul(z2, z))
            flat = [x, y, z] if g == 1 else [x[0], x[1], y[0], y[1], z[0], z[1]]
            jacs.append(orc.ints_to_limbs([P.fq_to_mont(v) for v in flat], 6).reshape(-1))
        jacs.append(np.zeros(3 * width, dtype=np.uint64))          # z == 0 -> infinity
        jac = np.vstack(jacs)
        n = jac.shape[0]
        aff = np.zeros((n, 2 * width), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        import ctypes as C
        rc = czk_amd.lib().czk_jac_to_affine(None, C.c_int(g), jac.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                             aff.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p))
        assert rc == 0
        for i in range(n):
            want, winf = orc.jac_to_affine(g, jac[i])
            assert bool(inf[i]) == winf and (winf or np.array_equal(aff[i], want)), (g, i)
        # czk_jac_add / czk_jac_add_mixed (host group law) against the checker, incl. doubling and infinity operands
        L = czk_amd.lib()
        pv = lambda a: a.ctypes.data_as(C.c_void_p)
        for i, j in ((0, 1), (2, 2), (3, n - 1), (n - 1, 4)):
            out = np.zeros(3 * width, dtype=np.uint64)
            assert L.czk_jac_add(None, C.c_int(g), pv(jac[i]), pv(jac[j]), pv(out)) == 0
            w1, i1 = orc.jac_to_affine(g, orc.jac_add(g, jac[i], jac[j]))
            w2, i2 = orc.jac_to_affine(g, out)
            assert i1 == i2 and (i1 or np.array_equal(w1, w2)), (g, i, j)
            a_aff, a_inf = orc.jac_to_affine(g, jac[j])
            out2 = np.zeros(3 * width, dtype=np.uint64)
            assert L.czk_jac_add_mixed(None, C.c_int(g), pv(jac[i]), pv(a_aff), C.c_int(int(a_inf)), pv(out2)) == 0
            w3, i3 = orc.jac_to_affine(g, out2)
            assert i1 == i3 and (i1 or np.array_equal(w1, w3)), (g, i, j)


def test_product_library_has_no_environment_switches_and_no_lab_kernels():
    """libczk_hip.so (what a prover links) imports no getenv and carries none of the measured-and-rejected kernel
    variants; libczk_hip_lab.so (tests / A/B tools) is the same source with -DCZK_LAB: same exported ABI, the variants and the
    environment-to-option shim included."""
    import os
    import subprocess
    import czk_amd
    pkg = os.path.dirname(czk_amd.lib_path())
    prod, lab = os.path.join(pkg, "libczk_hip.so"), os.path.join(pkg, "libczk_hip_lab.so")
    assert os.path.exists(lab), "build() makes both libraries"

    def symbols(path):
        return subprocess.run(["nm", "-D", "-C", path], capture_output=True, text=True, check=True).stdout
    p, q = symbols(prod), symbols(lab)
    assert " U getenv" not in p and " U secure_getenv" not in p
    assert " U getenv" in q
    lab_only = ("k_affine_round", "k_aff_build_first", "k_accumulate_u2p", "k_accumulate_u_lvl", "czk::k_accumulate<", "czk::k_reduce_level<")
    for k in lab_only:
        assert k not in p, f"{k} ships in the product library"
        assert k in q, f"{k} missing from the lab library"
    for k in ("k_accumulate_te", "k_accumulate_u2", "k_reduce_level_p", "k_ntt2"):
        assert k in p, k
    L = czk_amd.binding.lab_lib()
    assert [s for s in czk_amd.header_symbols() if not hasattr(L, s)] == []
    assert czk_amd.lib().czk_build_is_lab() == 0 and L.czk_build_is_lab() == 1
    # no product source reads the environment either
    csrc = os.path.join(pkg, "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            txt = open(os.path.join(csrc, fn)).read()
            if "getenv" in txt:
                assert fn == "core.hip" and txt.count("getenv(") == 1 and "#ifdef CZK_LAB" in txt, fn

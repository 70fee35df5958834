"""Builds libczk_hip.so (the product: HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libczk_hip.so")
# (source, extra flags).  msm.hip holds the setup / sort / reduction kernels and is built with the Montgomery
# multiply out of line (see field.h CZK_NOINLINE_MUL); the hot kernels keep it inlined.  core.hip is host code only (the host-side
# group operations): out of line too, 14 s instead of 4.5 min of host compilation.
SOURCES = [("core.hip", ["-DCZK_NOINLINE_MUL"]), ("lanes.hip", ["-DCZK_NOINLINE_MUL"]), ("ntt.hip", []), ("ntt_pass.hip", []), ("ntt_mixed.hip", []), ("msm.hip", ["-DCZK_NOINLINE_MUL"]), ("msm_acc_g1.hip", []),
           ("msm_acc_g2.hip", []), ("msm_red_g2.hip", []), ("msm_heavy_g2.hip", []), ("poly.hip", []), ("share.hip", [])]
HEADERS = ["field.h", "curve.h", "czk_internal.h", "msm_acc.h", "fq2p.h", "fq2pu.h", "fqu.h", "fqu_il.h", "fqu_mad_il.inc", "fru.h", "fru_constants.inc", "ntt_pass.h", "fq_safegcd.h", "msm_aff.h", "te.h", "te_constants.inc", os.path.join("..", "..", "include", "czk.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-Wno-unused-result", "-Wno-pass-failed"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        import time
        t0 = time.time()
        subprocess.check_call(cmd)
        if verbose:
            print("%6.1fs  %s" % (time.time() - t0, " ".join(cmd[-4:])), file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

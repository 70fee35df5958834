"""Builds the HIP libraries for gfx950 with hipcc, in-tree.

  libczk_hip.so      the PRODUCT: the kernels a prover runs + the C ABI of include/czk.h.  Reads no environment variables; tuning goes
                     through czk_ctx_set_option.
  libczk_hip_lab.so  the LAB build (-DCZK_LAB): the same sources plus the measured-and-rejected variants kept for A/B runs (csrc/lab/:
                     batched-affine rounds, safegcd inversion, interleaved multiply-add chains, Karatsuba Fq2; the lane-pair G2 accumulate
                     kernel, the saturated accumulate / reduction kernels) and the CZK_* environment switches of the measurement tools,
                     which it translates into options.  Same exported symbols.  Loaded by tests through Context(lab=True) and by tools
                     through CZK_LIB_PATH; never by the provers.

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
LIB_LAB = os.path.join(HERE, "libczk_hip_lab.so")
# (source, extra flags).  msm.hip holds the setup / sort / reduction kernels and is built with the Montgomery
# multiply out of line (see field.h CZK_NOINLINE_MUL); the hot kernels keep it inlined.  core.hip is host code only (the host-side
# group operations): out of line too, 14 s instead of 4.5 min of host compilation.
SOURCES = [("core.hip", ["-DCZK_NOINLINE_MUL"]), ("lanes.hip", ["-DCZK_NOINLINE_MUL"]), ("ntt.hip", []), ("ntt_pass.hip", []), ("ntt_mixed.hip", []), ("msm.hip", ["-DCZK_NOINLINE_MUL"]), ("msm_acc_g1.hip", []),
           ("msm_acc_g2.hip", []), ("msm_red_g2.hip", []), ("msm_heavy_g2.hip", []), ("poly.hip", []), ("share.hip", []), ("net.hip", ["-DCZK_NOINLINE_MUL"])]
HEADERS = ["field.h", "curve.h", "czk_internal.h", "msm_acc.h", "fq2p.h", "fq2pu.h", "fqu.h", "fru.h", "fru_constants.inc", "ntt_pass.h", "te.h", "te_constants.inc",
           os.path.join("..", "..", "include", "czk.h")]
LAB_HEADERS = [os.path.join("lab", h) for h in ("msm_aff.h", "fq_safegcd.h", "fqu_il.h", "fqu_mad_il.inc", "fq2u_karatsuba.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-pass-failed", "-I" + CSRC]
MAX_JOBS = int(os.environ.get("CZK_BUILD_JOBS", "0")) or max(4, (os.cpu_count() or 8))


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _plan(lab: bool, force: bool):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS + (LAB_HEADERS if lab else [])]
    objdir = os.path.join(CSRC, "obj_lab" if lab else "obj")
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + extra + (["-DCZK_LAB"] if lab else []) + ["-c", s, "-o", o])
    return objs, jobs


def build(force: bool = False, verbose: bool = False, lab: bool = True) -> str:
    """Builds the product library, and the lab library too unless lab=False.  Returns the product library's path."""
    import time

    def run(cmd):
        t0 = time.time()
        subprocess.check_call(cmd)
        if verbose:
            print("%6.1fs  %s" % (time.time() - t0, " ".join(cmd[-4:])), file=sys.stderr)

    targets = [(LIB, False)] + ([(LIB_LAB, True)] if lab else [])
    plans = [(lib, *_plan(is_lab, force)) for lib, is_lab in targets]
    jobs = [j for _, _, js in plans for j in js]
    if jobs:
        with ThreadPoolExecutor(max_workers=min(MAX_JOBS, len(jobs))) as ex:
            list(ex.map(run, jobs))
    for lib, objs, js in plans:
        if js or _stale(lib, objs):
            run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--no-lab" not in sys.argv))

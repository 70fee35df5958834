#!/usr/bin/env python3
"""bench.py -- collaborative Groth16 proofs/sec (BLS12-377, 2^20 constraints, SPDZ, 2 parties) on MI355X.

One "step" = the complete per-party local compute of ONE proof for BOTH parties of a 2-party SPDZ prover on one
GPU (BASELINE.json configs[1]): the R1CS->QAP witness map (7 share-vector NTT ops per party = 28 Fr NTT lanes of
D = 2^21, the Beaver local half and the pointwise steps) plus the 5 MSMs h / l / a / b_g1 / b_g2 for every share
lane (4 lanes: 2 parties x {sh, mac}), inputs already resident in HBM, outputs = the 20 MSM results on the host.

`--gpus N` runs N ranks, one per GPU: started WITHOUT a torchrun environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; started by torchrun it checks
that WORLD_SIZE == N.  Layout "replica" (default): every rank proves its own independent proofs (weak scaling, no
data-path collective: parties and proofs are independent units); value = N * steps /
max-over-ranks time.  Layout "party": ONE proof, party p on rank p (N == --parties), opens all-gathered over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (G1 bucket accumulation), timed live with
HIP events on the stream it runs on; `cpu_baseline` times the CPU checker (oracle/, the C restatement of the
reference algorithms) -- one thread on a bounded sample and all host cores; after the timed region every MSM result is
checked against [sum_i k_i s_i] G computed on the host from the known discrete logs of the bases (`results_checked`).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Hardware queues: a context uses 4 streams (the caller's + the MSM pipeline's sort / accumulate / reduce).  Groth16 runs ONE context: 8 queues.  The
# Plonk / Marlin workloads keep several proofs in flight, each on its own context: with 8 queues the streams of three or more contexts share
# queues and serialise (measured round 4: plonk 6.3 proofs/s with 3 in flight on 8 queues, 7.3 with 4 in flight on 24).  Must be set before HIP
# initialises, hence from argv.
_POLYIOP = any(a in ("plonk", "marlin") for a in sys.argv)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if _POLYIOP else "8")
if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
    # one node: rendezvous and bootstrap sockets on the loopback interface, no InfiniBand probing -- the ranks' data path is xGMI (RCCL) or
    # shared memory; without these gloo / RCCL enumerate interfaces and resolve the container's hostname first, which stalls for minutes where
    # the resolver times out (seen on the GPU boxes: "[c10d] The hostname of the client socket cannot be retrieved")
    for _k, _v in (("GLOO_SOCKET_IFNAME", "lo"), ("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1")):
        os.environ.setdefault(_k, _v)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
REF_PROOFS_PER_S = 3.0 / (328.957 + 317.213 + 320.422)   # reference's published 2^20 SPDZ-2pc timings
# csrc/fqu.h fqu_xyzz_acc_mixed: 6 multiplies (378 v_mad_u64_u32 each), 2 squarings (287), 1 fused two-product multiply (574)
MADS_PER_MIXED_ADD = 6 * 378 + 2 * 287 + (2 * 196 + 182)
# csrc/te.h teu_madd_s: G1 in twisted Edwards extended coordinates, 7 multiplies, no squarings
MADS_PER_MIXED_ADD_TE = 7 * 378
# csrc/fqu.h fq2u_xyzz_acc_mixed: 6 Fq2 products (2 x (2 x 196 + 182) each), 2 Fq2 squarings (2 Fq multiplies each), Y3 as two four-product sums
MADS_PER_MIXED_ADD_G2 = 6 * 1148 + 2 * 756 + 2 * (4 * 196 + 182)
# MI355X_MICROARCH.md: 256 CU x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles -> 256 x 4 x 32 x 2.4 GHz = 78.6 T lane-ops/s
# for a full-rate instruction.  v_mad_u64_u32 is NOT full rate: a register-only loop sustains 33.5 - 35.5 T lane-ops/s at a measured
# 2.42 GHz (tools/bank_bench.hip: 128 multiply-adds per asm statement, one dependent chain per lane or eight, any VGPR banks;
# profiles/r03_interleave.txt), i.e. 4.5 cycles per wave64 instruction -- that figure is the peak.  (Rounds 1 - 2 priced against 27.1 T from
# tools/rate_bench.hip, whose one-statement-per-instruction loops carry an s_nop per multiply-add; the better measurement lowers `frac`.)
VALU_FULL_RATE_GOPS = 256 * 4 * 32 * 2.4
MAD_PEAK_MEASURED_GOPS = 35000.0
NOMINAL_CLOCK_GHZ = 2.42
STREAM_ELAPSED_NOTE = ("HIP-event brackets on the library's internal streams, summed per stage: NOT a breakdown of the step -- the sort / accumulate / reduce / NTT "
                       "streams overlap, so the entries add up to more than ms_per_step.  Only msm_accumulate_g1 / _g2 bracket single kernels on a stream of their own; "
                       "accumulate_busy_ms_per_step is the union of those two")


def busy_union_ms(ctxs, names) -> float:
    """Time during which at least one kernel bracket named in `names` was running, over all contexts in `ctxs` (their profile
    origins merged onto the first context's clock): the union of the HIP-event intervals, NOT the sum of elapsed times."""
    iv = []
    for c in ctxs:
        off = 0.0 if c is ctxs[0] else ctxs[0].profile_base_offset(c)
        for name in names:
            a = c.profile_intervals(name)
            if len(a):
                iv.append(a + off)
    if not iv:
        return 0.0
    a = np.concatenate(iv)
    a = a[np.argsort(a[:, 0])]
    busy, cur0, cur1 = 0.0, a[0, 0], a[0, 1]
    for s0, s1 in a[1:]:
        if s0 > cur1:
            busy += cur1 - cur0
            cur0, cur1 = s0, s1
        else:
            cur1 = max(cur1, s1)
    return float(busy + cur1 - cur0)


class PowerSampler:
    """Board power of one GPU over a timed region: a thread reads the amdgpu hwmon file (microwatts; a file read, ~20 us) every 10 ms.  Both dominant
    kernels run at the board's power limit, so proofs per joule is the axis that tells whether a change traded clock for
    instructions.  Falls back to polling `rocm-smi --showpower` (a few samples per second) where sysfs has no power file; report() returns None
    fields when neither works.  Reporting only: never raises."""

    def __init__(self, device: int):
        import glob
        import threading
        self.samples, self.source, self._stop, self._thread = [], None, threading.Event(), None
        try:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"), key=lambda f: int(f.split("/card")[1].split("/")[0]))
            cards = [c for c in cards if any(os.path.exists(os.path.join(c, f)) for f in ("power1_average", "power1_input"))]
            pick = None
            try:                                    # the card whose PCI address is this device's
                import torch
                pr = torch.cuda.get_device_properties(device)
                want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
                for c in cards:
                    if want in os.path.realpath(os.path.join(c, "..", "..")):
                        pick = c
            except Exception:      # noqa: BLE001
                pick = None
            if pick is None and cards:
                pick = cards[min(device, len(cards) - 1)]
            if pick is not None:
                self._file = next(os.path.join(pick, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(pick, f)))
                float(open(self._file).read())
                self.source = "sysfs " + self._file
        except Exception:      # noqa: BLE001
            self.source = None
        if self.source is None:
            import shutil
            if shutil.which("rocm-smi"):
                self.source = "rocm-smi --showpower"
        self._device = device

    def _read(self):
        if self.source and self.source.startswith("sysfs"):
            return float(open(self._file).read()) / 1e6
        import re
        out = subprocess.run(["rocm-smi", "-d", str(self._device), "--showpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Power \(W\): ([0-9.]+)", out)
        return float(m.group(1)) if m else None

    def _loop(self):
        period = 0.01 if self.source.startswith("sysfs") else 0.05
        while not self._stop.is_set():
            try:
                w = self._read()
                if w is not None:
                    self.samples.append(w)
            except Exception:      # noqa: BLE001
                pass
            self._stop.wait(period)

    def start(self):
        if self.source is None:
            return self
        import threading
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=15)
        return self

    def report(self, proofs: float, seconds: float) -> dict:
        if not self.samples:
            return {"power_w_avg": None, "proofs_per_kJ": None, "power_source": self.source, "power_samples": 0}
        avg = float(sum(self.samples) / len(self.samples))
        return {"power_w_avg": avg, "power_w_max": float(max(self.samples)), "proofs_per_kJ": proofs / (avg * seconds / 1e3) if avg > 0 and seconds > 0 else None,
                "joules_per_proof": avg * seconds / proofs if proofs else None, "power_source": self.source, "power_samples": len(self.samples)}


def csrc_digest() -> str:
    """SHA-256 over the product's kernel sources (csrc/*.hip, *.h, *.inc; not csrc/lab/) in name order: stored with a PMC profile
    (tools/pmc_summary.py) and recomputed here, so that a bench line can say whether the kernels it ran are the ones the counters were
    taken from -- without git history (the GPU box has none)."""
    d = os.path.join(ROOT, "collaborative-zksnark_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".inc")) and fn != "net.hip":     # net.hip is the communicator: host code, no kernel of its own
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


# configs[0] is the reference's own CPU-runnable case: `./scripts/bench.zsh groth16 spdz 10 2` = 10 squarings (mpc-snarks/scripts/bench.zsh:23,55 passes
# --computation-size 10 -> 10 constraints, FFT domain 16: mpc-snarks/src/proof.rs:476-477), and BASELINE.json words it as 2^10: both are reported.  Their metric is the
# reference's own -- ONE proof's wall time (the "timed section", proof.rs:130-139) -- so `latency_ms_single_proof` is the figure; the reference's published times
# beside them are from mpc-snarks/analysis/data/mpc.csv (rows N,groth16,spdz,1,2), other hardware.
CONFIG0_REFERENCE_S = {"groth16_spdz2_10": {"constraints_8": 0.036529, "constraints_16": 0.06793, "source": "mpc-snarks/analysis/data/mpc.csv rows 8 / 16,groth16,spdz,1,2 (N = 10 lies between)"},
                       "groth16_spdz2_2e10": {"constraints_1024": 0.693, "source": "mpc-snarks/analysis/data/mpc.csv row 1024,groth16,spdz,1,2 (BASELINE.md section 1)"}}
OTHER_WORKLOADS = (   # (key, bench.py arguments, HBM the run needs in GB) -- BASELINE configs[0], [2], [3] and the configs[4] size on ONE GPU
    ("groth16_spdz2_10", ["--workload", "groth16", "--parties", "2", "--constraints", "10", "--steps", "20", "--warmup", "3", "--no-seam-report", "--verify-report"], 2),
    ("groth16_spdz2_2e10", ["--workload", "groth16", "--parties", "2", "--log-n", "10", "--steps", "20", "--warmup", "3", "--no-seam-report", "--verify-report"], 2),
    ("plonk_gsz3_2e18", ["--workload", "plonk", "--parties", "3", "--log-n", "18", "--steps", "12", "--warmup", "4"], 20),
    ("marlin_spdz2_2e20", ["--workload", "marlin", "--parties", "2", "--log-n", "20", "--steps", "12", "--warmup", "4"], 40),
    ("groth16_spdz2_2e22", ["--workload", "groth16", "--parties", "2", "--log-n", "22", "--steps", "4", "--warmup", "2", "--no-seam-report"], 70),
)


def other_workloads_report(device: int) -> dict:
    """The other BASELINE configs' workloads on this GPU, each as its own short `bench.py` process (own context, own key): what the
    driver's default run would otherwise never see.  Never `value`."""
    import torch
    out = {}
    for key, argv, need_gb in OTHER_WORKLOADS:
        free_gb = torch.cuda.mem_get_info(device)[0] / 2**30
        if free_gb < need_gb:
            out[key] = {"skipped": f"{free_gb:.0f} GB of HBM free, {need_gb} GB needed"}
            continue
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--device", str(device), "--no-cpu-baseline", "--no-other-workloads"] + argv,
                               capture_output=True, text=True, timeout=600,
                               env={**{k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}, "CZK_BENCH_CHILD": "1"})
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[key] = {"error": (r.stdout + r.stderr)[-400:]}
                continue
            j = json.loads(line[-1])
            out[key] = {"proofs_per_s": j["value"], "ms_per_proof": j["ms_per_step"], "results_checked": bool(j.get("results_checked")),
                        "accumulate_busy_frac": j.get("accumulate_busy_frac"), "steps": j["steps"], "metric": j["metric"],
                        "proofs_in_flight": j["config"].get("proofs_in_flight", "pipelined"), "wall_s": time.time() - t0,
                        "power_w_avg": j.get("power_w_avg"), "proofs_per_kJ": j.get("proofs_per_kJ"),
                        # the same roofline block as the headline, for this workload's dominant kernel (HIP-event timed inside the child)
                        "roofline": {k: (j.get("roofline") or {}).get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches")},
                        "command": "python bench.py " + " ".join(argv)}
            if j.get("proof_verifies") is not None:
                out[key]["proof_verifies"] = j["proof_verifies"]
            if key in CONFIG0_REFERENCE_S:
                # the reference's metric for this config: one proof's wall time, inputs resident (first_proof_ms also builds the NTT tables / sizes the workspaces)
                out[key].update({"latency_ms_single_proof": j.get("latency_ms_single_proof"), "first_proof_ms": j.get("first_proof_ms"), "one_shot_s": j.get("one_shot_s"),
                                 "constraints": j["config"].get("constraints"), "domain": j["config"].get("domain"), "results_sha256": j["config"].get("results_sha256"),
                                 "reference_published_s": CONFIG0_REFERENCE_S[key]})
            if argv[1] in ("plonk", "marlin"):
                # the same workload from the compiled host (tools/polyvm_host.hpp over include/czk.h: no torch, no Python, one device arena)
                try:
                    hd = [host_demo_exe(), argv[1]] + argv[2:]
                    r2 = subprocess.run(hd, capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
                    l2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
                    j2 = json.loads(l2[-1]) if r2.returncode == 0 and l2 else None
                    out[key]["cpp_host"] = ({k: j2[k] for k in ("harness", "proofs_per_s", "ms_per_proof", "latency_ms_single_proof", "proofs_in_flight", "arena_peak_gb",
                                                                "in_flight_provers_equal", "output_sha256")} | {"fraction_of_python_host": j2["proofs_per_s"] / j["value"]}) \
                        if j2 else {"error": (r2.stdout + r2.stderr)[-300:]}
                except Exception as e:      # noqa: BLE001
                    out[key]["cpp_host"] = {"error": repr(e)[-300:]}
        except Exception as e:      # noqa: BLE001 -- the report must not take the headline down with it
            out[key] = {"error": repr(e)[-400:]}
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(n: int, argv: list[str]) -> int:
    """Re-executes this script as n ranks (one process per GPU) on this node and returns the job's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=dict(os.environ, CZK_BENCH_CHILD="1"))


def per_rank_report(parallel, dt_local: float, steps: int, device, power: dict | None = None) -> list:
    """[{rank, ms_per_step, sclk_mhz_by_card}] of every rank (before the max over ranks): shows a straggler GPU or a clock that power management
    holds lower on one device.  The clocks are a single sample right after the timed region (None when sysfs does not expose them)."""
    import torch
    rank, world, _ = parallel.env_rank_world()
    clk = None
    try:
        # current shader clock of every card, from sysfs (the line marked `*` of pp_dpm_sclk): a file read -- the SMI libraries behind
        # torch.cuda.clock_rate take tens of seconds to initialise in a fresh process
        import glob
        clk = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            cur = [ln for ln in open(f).read().splitlines() if ln.rstrip().endswith("*")]
            clk.append(int(cur[0].split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")) if cur else None)
    except Exception:      # noqa: BLE001 -- reporting only
        clk = None
    mine = {"rank": rank, "ms_per_step": dt_local / max(1, steps) * 1e3, "sclk_mhz_by_card": clk}
    if power:
        mine.update({k: power.get(k) for k in ("power_w_avg", "proofs_per_kJ")})
    if world == 1 or not torch.distributed.is_initialized():
        return [mine]
    got = [None] * world
    torch.distributed.all_gather_object(got, mine)
    return got


# What `bench.py --gpus N` adds to its JSON line when it runs on N > 1 GPUs (the driver's SCALE run): after the replica line, rank 0 runs
# these as short children over the same N GPUs.  Every entry is one open question of DESIGN.md section 6; see multi_gpu_report.
_EXCH = (("ring", "torch"), ("p2p", "torch"), ("ring", "czk"), ("p2p", "czk"))


def multi_gpu_plan(n: int) -> list:
    """[(key, argv of a bench.py child, reference key or None)]: the party layout of the BASELINE config whose party count is n under both
    exchange patterns and both transports, its one-GPU layout as the digest reference, and the split layout of the headline."""
    g16_20 = ["--workload", "groth16", "--log-n", "20", "--steps", "4", "--warmup", "1", "--no-seam-report"]
    cfgs = {
        2: [("marlin_spdz2_2e20", ["--workload", "marlin", "--parties", "2", "--log-n", "20", "--steps", "4", "--warmup", "2"]),       # configs[3]
            ("groth16_spdz2_2e20", g16_20 + ["--parties", "2"])],
        3: [("plonk_gsz3_2e18", ["--workload", "plonk", "--parties", "3", "--log-n", "18", "--steps", "4", "--warmup", "2"]),            # configs[2]
            ("groth16_spdz3_2e20", g16_20 + ["--parties", "3"])],
        4: [("groth16_spdz4_2e20", g16_20 + ["--parties", "4"])],
        8: [("groth16_spdz8_2e22", ["--workload", "groth16", "--parties", "8", "--log-n", "22", "--steps", "3", "--warmup", "1", "--no-seam-report"]),   # configs[4]
            ("groth16_spdz8_2e22_no_tables", ["--workload", "groth16", "--parties", "8", "--log-n", "22", "--steps", "3", "--warmup", "1", "--no-seam-report",
                                              "--no-tables"])],
    }.get(n, [(f"groth16_spdz{n}_2e20", g16_20 + ["--parties", str(n)])])
    # Most decisive first (the budget may cut the list short): the party layout through the library's own communicator, then through torch.distributed,
    # ring before p2p; then the one-GPU reference runs that the digests are compared with (a party digest that equals the other party digests of
    # its config is already strong evidence); then the split layout.
    plan = []

    def party(key, argv, net, exch):
        return (f"{key}/party/{net}/{exch}", ["--gpus", str(n), "--layout", "party", "--exchange", exch, "--net", net] + argv, key + "/one_gpu")
    for net, exch in (("czk", "ring"), ("torch", "ring"), ("czk", "p2p"), ("torch", "p2p")):
        plan += [party(key, argv, net, exch) for key, argv in cfgs if not key.endswith("_no_tables")]
        if (net, exch) == ("czk", "ring"):
            # ... and, right after the first party runs, the Groth16 party layout under a REAL key: the parties' proof shares are gathered and the opened proof
            # goes through the verification equation (`proof_verifies`) -- the reference's own acceptance test, over its own layout and over RCCL
            plan += [(f"{key}/verify/czk/ring", ["--gpus", str(n), "--layout", "party", "--exchange", "ring", "--net", "czk", "--real-key"] + argv, None)
                     for key, argv in cfgs if key.startswith("groth16") and not key.endswith("_no_tables")]
    plan.append(("groth16_spdz2_2e20/split", ["--gpus", str(n), "--layout", "split", "--parties", "2"] + g16_20, "headline"))
    # the library's third transport between GPUs: device mailboxes mapped into the peers with hipIpc (peer access over xGMI), no RCCL involved
    plan += [party(key, argv, "czk-ipc", "ring") for key, argv in cfgs if not key.endswith("_no_tables")]
    for net in ("czk", "torch"):   # with vs without window tables: after the transport / pattern questions
        plan += [party(key, argv, net, "ring") for key, argv in cfgs if key.endswith("_no_tables")]
    for key, argv in cfgs:
        one_gpu = argv + (["--no-tables"] if "--log-n" in argv and argv[argv.index("--log-n") + 1] == "22" and "--no-tables" not in argv else [])
        plan.append((key + "/one_gpu", ["--gpus", "1", "--inflight", "1"] + one_gpu, None))   # all parties' lanes on ONE GPU: the digest every party layout must reproduce
    return plan


def _run_child(cmd, env, timeout):
    """subprocess.run with the child in its own process group, so that a timeout takes the launcher, its torchrun and every rank down together
    (an orphaned rank would keep its GPU into the next run)"""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)       # exactly the group this call created
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        return 124, out, err + "\n[bench.py] child exceeded its time budget and was killed with its process group"
    return p.returncode, out, err


def multi_gpu_report(n: int, headline: dict, dry_run: bool, budget_s: float) -> dict:
    """Runs multi_gpu_plan(n) as children (rank 0 only; the other ranks of the replica run have exited and freed their GPUs) and reduces
    each child's JSON line to the fields that answer DESIGN.md section 6's open questions:
      value / ms_per_step / latency_ms_single_proof   ring vs p2p, torch.distributed vs czk_net, with vs without tables, split latency vs one GPU
      backend / net / ranks_seen_by_backend              that the RCCL branch really ran with n ranks
      per_rank                                           a straggler GPU, per-device clocks
      digest_equals_reference                            the layout reproduces the one-GPU layout's proof elements"""
    scrub = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
             "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES", "OMP_NUM_THREADS", "CZK_BENCH_CHILD")
    env = {k: v for k, v in os.environ.items() if k not in scrub and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_ASYNC"))}
    env["CZK_BENCH_REPORT_CHILD"] = "1"
    out, digests, t_start = {}, {"headline": (headline or {}).get("config", {}).get("results_sha256")}, time.time()
    exercised = set()
    for key, argv, ref in multi_gpu_plan(n):
        kind = key.split("/")[1]                     # one_gpu / party / split
        if dry_run and kind in exercised:            # CPU test of the report's shape: one launch per kind of child, the rest as planned commands
            out[key] = {"dry_run": "planned", "reference": ref, "command": "python bench.py " + " ".join(argv)}
            continue
        exercised.add(kind)
        left = budget_s - (time.time() - t_start)
        if left < 30:
            out[key] = {"skipped": f"report budget of {budget_s:.0f} s used up"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-workloads", "--no-multi-gpu-report", "--no-verify-report"] + argv + (["--dry-run"] if dry_run else [])
        t0 = time.time()
        try:
            rc, c_out, c_err = _run_child(cmd, env, min(420.0, left))
            line = [ln for ln in c_out.splitlines() if ln.startswith("{")]
            if rc != 0 or not line:
                out[key] = {"error": (c_out + c_err)[-400:], "command": "python bench.py " + " ".join(argv)}
                continue
            j = json.loads(line[-1])
            if dry_run:
                out[key] = {"dry_run": True, "n_gpus": j.get("n_gpus"), "ranks_seen_by_backend": j.get("ranks_seen_by_backend"), "layout": j.get("layout"),
                            "reference": ref, "command": "python bench.py " + " ".join(argv)}
                continue
            dg = j.get("config", {}).get("results_sha256")
            digests[key] = dg
            out[key] = {"proofs_per_s": j["value"], "ms_per_proof": j["ms_per_step"], "latency_ms_single_proof": j.get("latency_ms_single_proof"),
                        "n_gpus": j["n_gpus"], "ranks_seen_by_backend": j.get("ranks_seen_by_backend"), "backend": j.get("backend"), "net": j.get("net"),
                        "layout": j["config"].get("layout"), "exchange": argv[argv.index("--exchange") + 1] if "--exchange" in argv else None,
                        "per_rank": j.get("per_rank"), "results_checked": bool(j.get("results_checked")), "results_sha256": dg,
                        **({"proof_verifies": j["proof_verifies"]} if j.get("proof_verifies") is not None else {}),
                        "reference": ref, "wall_s": time.time() - t0, "command": "python bench.py " + " ".join(argv)}
        except Exception as e:      # noqa: BLE001 -- the report must not take the replica line down with it
            out[key] = {"error": repr(e)[-400:], "command": "python bench.py " + " ".join(argv)}
    # digests: against the one-GPU reference where it ran, and among the layouts of one config in any case
    for key, v in out.items():
        if "results_sha256" not in v:
            continue
        ref = v.get("reference")
        v["digest_equals_reference"] = (v["results_sha256"] == digests[ref]) if ref and digests.get(ref) else None
        peers = [w["results_sha256"] for k2, w in out.items() if k2 != key and k2.split("/")[0] == key.split("/")[0] and "/party/" in k2 and "results_sha256" in w]
        v["digest_equals_other_party_runs"] = all(d == v["results_sha256"] for d in peers) if peers and "/party/" in key else None
    return out


# ---------------------------------------------------------------------------------------------------------------
# host-side result check: pure-Python big-integer group arithmetic (no library, no checker code involved)
# ---------------------------------------------------------------------------------------------------------------
class _Fq:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q_MOD)
    sub = staticmethod(lambda a, b: (a - b) % Q_MOD)
    mul = staticmethod(lambda a, b: a * b % Q_MOD)
    inv = staticmethod(lambda a: pow(a, Q_MOD - 2, Q_MOD))


class _Fq2:   # Fq[u] / (u^2 + 5)  (curves/bls12_377/src/fields/fq2.rs:13)
    zero, one = (0, 0), (1, 0)
    add = staticmethod(lambda a, b: ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD))
    sub = staticmethod(lambda a, b: ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD))
    mul = staticmethod(lambda a, b: ((a[0] * b[0] - 5 * a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD))

    @staticmethod
    def inv(a):
        n = pow((a[0] * a[0] + 5 * a[1] * a[1]) % Q_MOD, Q_MOD - 2, Q_MOD)
        return (a[0] * n % Q_MOD, -a[1] * n % Q_MOD)


def _ec_scalar_mul(F, gen, e: int):
    """[e] gen on y^2 = x^3 + b (a = 0) with affine big-integer arithmetic; returns None for infinity."""
    def add(p, q):
        if p is None:
            return q
        if q is None:
            return p
        if p[0] == q[0]:
            if F.add(p[1], q[1]) == F.zero:
                return None
            lam = F.mul(F.mul(F.add(F.add(p[0], p[0]), p[0]), p[0]), F.inv(F.add(p[1], p[1])))   # 3 x^2 / 2 y
        else:
            lam = F.mul(F.sub(q[1], p[1]), F.inv(F.sub(q[0], p[0])))
        x = F.sub(F.sub(F.mul(lam, lam), p[0]), q[0])
        return (x, F.sub(F.mul(lam, F.sub(p[0], x)), p[1]))
    acc = None
    for bit in bin(e)[2:] if e else "":
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, gen)
    return acc


def _dot_mod_r(k: np.ndarray, s: np.ndarray) -> int:
    """sum_i k_i * s_i mod r for (n,4) uint64 limb arrays, exactly: 16-bit limbs as float64, 16 x 16 column sums by
    dgemm in chunks small enough (2^18 rows: < 2^50) to stay exact, accumulated as Python integers."""
    tot = [[0] * 16 for _ in range(16)]
    n = k.shape[0]
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        K = np.ascontiguousarray(k[lo:hi]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        S = np.ascontiguousarray(s[lo:hi]).view(np.uint16).reshape(-1, 16).astype(np.float64)
        m = K.T @ S
        for a in range(16):
            for b in range(16):
                tot[a][b] += int(m[a, b])
    return sum(tot[a][b] << (16 * (a + b)) for a in range(16) for b in range(16)) % R_MOD


def check_results(czk, ctx, prover, results) -> dict:
    """Every lane of every MSM result == [sum_i k_i s_i mod r] G, with k_i the known discrete logs of the bases (seeds in
    provers.py), s_i the scalars the MSM consumed (copied back from HBM) and G the generator -- all on the host."""
    from czk_amd.provers import BASE_SEED, QUERIES, rand_fr_canonical
    t0 = time.perf_counter()
    r_inv = pow(1 << 256, -1, R_MOD)
    q_rinv = pow(1 << 384, -1, Q_MOD)

    def fq_ints(limbs):   # Montgomery limbs -> canonical python ints, 6 u64 each
        flat = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 6)
        return [sum(int(v[j]) << (64 * j) for j in range(6)) * q_rinv % Q_MOD for v in flat]
    one = np.array([[1, 0, 0, 0]], dtype=np.uint64)
    g1 = fq_ints(ctx.fixed_base_points(czk.CZK_G1, one))            # [1] G (the generator itself is KAT-pinned in tests/)
    g2 = fq_ints(ctx.fixed_base_points(czk.CZK_G2, one))
    gens = {1: (g1[0], g1[1]), 2: ((g2[0], g2[1]), (g2[2], g2[3]))}
    scal = prover.msm_scalars()
    checked = 0
    for name, sd in QUERIES:
        g = 2 if name == "b_g2" else 1
        n = prover.query_len[name]
        k = rand_fr_canonical(BASE_SEED + sd, n)
        if name.startswith("b_"):
            k[0] = 0                                                 # that base is the point at infinity
        s_all = scal[name].cpu().numpy().view(np.uint64)
        aff, inf = ctx.jac_to_affine(czk.CZK_G2 if g == 2 else czk.CZK_G1, results[name])
        for ln in range(prover.lanes):
            e = _dot_mod_r(k, s_all[ln, :n]) * r_inv % R_MOD         # the scalars are Montgomery: s_i = s_i_mont / R
            want = _ec_scalar_mul(_Fq2 if g == 2 else _Fq, gens[g], e)
            if want is None:
                assert inf[ln], f"{name} lane {ln}: expected infinity"
            else:
                got = fq_ints(aff[ln])
                got = (got[0], got[1]) if g == 1 else ((got[0], got[1]), (got[2], got[3]))
                assert not inf[ln] and got == want, f"{name} lane {ln}: MSM result differs from [sum k_i s_i] G"
            checked += 1
    return {"results_checked": True, "results_checked_points": checked, "results_check_s": round(time.perf_counter() - t0, 2)}


def _ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return [sum(int(v[j]) << (64 * j) for j in range(a.shape[-1])) for v in a.reshape(-1, a.shape[-1])]


def real_key_for(n_constraints: int):
    """(key, key scalars for Groth16Local, (r, s) limbs, r, s): a REAL Groth16 key of the benchmark circuit from known toxic waste (tests/groth16_real_key.py)"""
    from groth16_real_key import key_scalars, real_key
    from czk_amd.provers import rand_fr_canonical
    key = real_key(n_constraints, _ints(rand_fr_canonical(0x7A11 + n_constraints, 5)))
    rs = rand_fr_canonical(0xC0FFEE + 77 + n_constraints, 2)
    r, s = _ints(rs)
    return key, key_scalars(key), rs, r, s


def open_proof_shares(czk, ctx, shares, scheme: str, parties: int):
    """reveal: the parties' sh-lane shares of Proof{a, b, c} (Jacobian limbs, one per party) added up (x 1 / n for Shamir shares on the n-th roots of unity)"""
    G = {"a": czk.CZK_G1, "b": czk.CZK_G2, "c": czk.CZK_G1}
    ninv = pow(parties, -1, R_MOD) if scheme == "gsz" else 1
    opened = {}
    for k, g in G.items():
        acc = shares[k][0]
        for sh in shares[k][1:]:
            acc = ctx.jac_add(g, acc, sh)
        if scheme == "gsz":
            acc = ctx.jac_scalar_mul(g, acc, np.array([(ninv >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64))
        opened[k] = ctx.jac_to_affine(g, acc)
    return opened, ninv


def verify_opened_proof(czk, ctx, key, r: int, s: int, opened, h_acc: int) -> dict:
    """The opened proof against (1) [a] G1, [b] G2, [c] G1 for the exponents the prover equations give (plain witness, r, s, h_acc = the h MSM in the exponent) and
    (2) the verification equation in the exponent, e(A, B) = e(alpha, beta) e(IC, gamma) e(C, delta); big-integer Python only."""
    from groth16_real_key import expected_exponents
    from czk_amd.provers import rand_fr_canonical
    G = {"a": czk.CZK_G1, "b": czk.CZK_G2, "c": czk.CZK_G1}
    w0 = _ints(rand_fr_canonical(0xC0FFEE, 1))[0]
    a_exp, b_exp, c_exp, verifies, qap = expected_exponents(key, w0, r, s, h_acc)
    q_rinv = pow(1 << 384, -1, Q_MOD)
    fq_ints = lambda limbs: [v * q_rinv % Q_MOD for v in _ints(np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 6))]   # noqa: E731
    one = np.array([[1, 0, 0, 0]], dtype=np.uint64)
    g1, g2 = fq_ints(ctx.fixed_base_points(czk.CZK_G1, one)), fq_ints(ctx.fixed_base_points(czk.CZK_G2, one))
    gens = {czk.CZK_G1: (g1[0], g1[1]), czk.CZK_G2: ((g2[0], g2[1]), (g2[2], g2[3]))}
    points_ok = True
    for k, e in (("a", a_exp), ("b", b_exp), ("c", c_exp)):
        g = G[k]
        want = _ec_scalar_mul(_Fq2 if g == czk.CZK_G2 else _Fq, gens[g], e)
        aff, inf = opened[k]
        got = fq_ints(aff[0])
        got = (got[0], got[1]) if g == czk.CZK_G1 else ((got[0], got[1]), (got[2], got[3]))
        points_ok = points_ok and want is not None and not inf[0] and got == want
    return {"proof_verifies": bool(points_ok and verifies and qap), "proof_elements_match_prover_equations": bool(points_ok), "verification_equation_holds": bool(verifies),
            "qap_identity_holds": bool(qap)}


def verify_report(czk, torch, device, tstream, n_constraints: int, parties: int, scheme: str) -> dict:
    """One proof of THIS configuration under a REAL proving key, checked against the Groth16 verification equation -- the reference's own acceptance
    criterion (mpc-snarks/src/proof.rs:140-143 asserts verify_proof).  The timed runs use a synthetic key (random points), whose proofs cannot verify;
    here the key's discrete logs are generated from known toxic waste (tests/groth16_real_key.py: groth16/src/generator.rs restated on integers, the
    domain generator from the reference's constants), the GPU builds the points and proves -- constraint evaluation, witness map with both opens, the five
    MSMs on every share lane, create_proof's group steps -- and the opened proof must (1) equal [a] G1, [b] G2, [c] G1 for the exponents the prover
    equations give from the plain witness and the quotient h, and (2) satisfy a b = alpha beta + (sum x_i gamma_abc_i) gamma + c delta (mod r): e(A, B) =
    e(alpha, beta) e(IC, gamma) e(C, delta) in the exponent.  Host side: big-integer Python only (no library arithmetic, no checker code)."""
    from groth16_real_key import R_INV
    from czk_amd.provers import Groth16Local
    t0 = time.perf_counter()
    key, ks, rs, r, s = real_key_for(n_constraints)
    t_key = time.perf_counter() - t0
    ctx = czk.Context(device, tstream.cuda_stream)
    p = Groth16Local(czk, ctx, n_constraints, parties, scheme=scheme, key_scalars=ks)
    p.step()
    proof = p.create_proof({k: v.copy() for k, v in p.results.items()}, rs[0], rs[1])
    opened, ninv = open_proof_shares(czk, ctx, {k: [proof[k][p.lpp * j] for j in range(parties)] for k in "abc"}, scheme, parties)
    h_lanes = p.ab.cpu().numpy().view(np.uint64)
    D = key["D"]
    h_acc = sum(_dot_mod_r(h_lanes[p.lpp * j][:D - 1], ks["h"]) for j in range(parties)) * R_INV % R_MOD * ninv % R_MOD
    res = verify_opened_proof(czk, ctx, key, r, s, opened, h_acc)
    del p
    ctx.close()
    torch.cuda.empty_cache()
    assert res["proof_verifies"], res
    return {**res, "constraints": n_constraints, "parties": parties,
            "scheme": scheme, "key": "real: discrete logs from known toxic waste (tests/groth16_real_key.py), points built by czk_fixed_base_points",
            "key_generation_s": round(t_key, 2), "seconds": round(time.perf_counter() - t0, 2),
            "note": "e(A, B) = e(alpha, beta) e(sum x_i gamma_abc_i, gamma) e(C, delta) checked in the exponent (all discrete logs known): the proof of this configuration "
                    "under a real key verifies; the timed runs use a synthetic key"}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (the only leg that touches oracle/)
# ---------------------------------------------------------------------------------------------------------------
def _ref_work(n_constraints: int) -> float:
    """Fq-multiplication-equivalents of the REFERENCE algorithm for one share lane at N constraints:
    Pippenger with c = ceil(log2 n)*69/100 + 2 (variable_base.rs:21-25): ceil(253/c) windows x (n mixed adds (11 M) +
    2*(2^c - 1) full adds (16 M)); G2 costs 3x; 7 NTTs of D log2(D)/2 Fr multiplications (one Fr mul ~ 0.45 Fq mul)."""
    N = n_constraints
    D = 1 << (N + 1).bit_length()

    def msm(n, g2=False):
        lg = (n - 1).bit_length() if n & (n - 1) else n.bit_length() - 1
        c = 3 if n < 32 else lg * 69 // 100 + 2
        w = -(-253 // c)
        return w * (n * 11 + 2 * ((1 << c) - 1) * 16) * (3 if g2 else 1)
    ntt = 7 * D * (D.bit_length() - 1) / 2 * 0.45
    return msm(D - 1) + msm(N) + 2 * msm(N + 1) + msm(N + 1, True) + ntt


def cpu_baseline(log_n_sample: int, log_n_full: int, parties: int, all_cores_log_n: int | None = None):
    """(i) ONE host thread -- the reference's published configuration (no `parallel` feature) -- on a bounded sample:
    oracle/libczk_oracle.so (limb-exact C restatement of the reference algorithms: same Pippenger window rule, same
    io/oi FFT) runs one proof's local compute for all share lanes at 2^log_n_sample constraints; the figure is scaled to
    the full size by the reference algorithm's multiplication count (model validated at 2^12..2^18 and one full 2^20 run:
    profiles/r02_cpu_baseline_validation.json).  (ii) ALL host cores, OpenMP tasks over MSM windows (the reference's
    `parallel` feature does the same with rayon), NTT butterflies, lanes and MSMs: a real timing at full size when the
    host has >= 64 hardware threads, otherwise at 2^16 and scaled."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from util import rand_fr_canonical
    lanes = 2 * parties
    cores = os.cpu_count() or 1

    def inputs(log_n):
        N = 1 << log_n
        log_d = (N + 1).bit_length()
        D = 1 << log_d
        b1 = orc.chain_points(1, D)       # distinct subgroup points: P_{i+1} = 2 P_i + G (timing does not depend on the values)
        b2 = orc.chain_points(2, N + 1)
        x = orc.fr_from_repr(rand_fr_canonical(5, D))
        return N, log_d, D, b1, b2, x
    N, log_d, D, b1, b2, x = inputs(log_n_sample)
    inf = np.zeros(D, dtype=np.uint8)
    t0 = time.perf_counter()
    for _ in range(lanes):
        a, b = orc.witness_map_pre(x, x, log_d)
        ab = orc.fr_mul(a, b)                      # stands in for the Beaver local half (same op count order)
        h = orc.witness_map_post(ab, x, log_d)
        orc.multi_scalar_mul(1, b1[:D - 1], inf, h)
        orc.multi_scalar_mul(1, b1[:N], inf, x[:N])
        orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1])
        orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1])
        orc.multi_scalar_mul(2, b2[:N + 1], inf, x[:N + 1])
    dt = time.perf_counter() - t0
    scale = _ref_work(1 << log_n_full) / _ref_work(1 << log_n_sample)
    cal, cal_src, full_s = 1.0, None, None   # measured (one full single-thread run at 2^20 on the GPU box's host) / (model prediction from the 2^14 sample)
    if log_n_sample == 14 and log_n_full == 20:
        for name in ("r06_cpu_baseline_validation.json", "r05_cpu_baseline_validation.json", "r02_cpu_baseline_validation.json"):     # the newest validation run wins
            try:
                v = json.load(open(os.path.join(ROOT, "profiles", name)))
                cal, cal_src, full_s = float(v["calibration_2^14_to_2^20"]), "profiles/" + name, v.get("single_thread_2^20_seconds", 513.9)
                break
            except Exception:      # noqa: BLE001
                continue
    out = {"value": 1.0 / (dt * scale * cal), "unit": "proofs/s", "cores": 1, "kind": "port",
           "sample": f"oracle C restatement, 1 thread: full local compute of one proof ({lanes} share lanes: witness map + 5 MSMs each) "
                     f"at 2^{log_n_sample} constraints took {dt:.2f} s; scaled x{scale:.1f} to 2^{log_n_full} by the reference algorithm's "
                     f"field-multiplication count and x{cal:.3f} by the measured error of that model at full size"
                     + (f" (one complete single-thread 2^20 run on this host class: {full_s:.1f} s; {cal_src})" if cal_src else "")}
    # (ii) all host cores
    if all_cores_log_n is None:
        all_cores_log_n = log_n_full if cores >= 64 else min(log_n_full, 16)
    if all_cores_log_n != log_n_sample:
        N, log_d, D, b1, b2, x = inputs(all_cores_log_n)
    xs = np.ascontiguousarray(np.broadcast_to(x, (lanes,) + x.shape))
    inf_b = np.zeros(N + 1, dtype=np.uint8)
    bufs = [xs.copy(), xs.copy(), xs.copy(), xs[:, :N].copy(), xs[:, :N + 1].copy()]
    threads = min(orc.max_threads(), 32)   # measured on the GPU box's host: 32 threads 35.0 s, 64: 37.1 s, 256: 60.3 s (container CPU limits)
    t0 = time.perf_counter()
    orc.groth16_local_par(log_d, N, *bufs, b1[:D - 1], b1[:N], b1[:N + 1], b1[:N + 1], b2, inf_b, threads=threads)
    dt_mt = time.perf_counter() - t0
    scale_mt = _ref_work(1 << log_n_full) / _ref_work(1 << all_cores_log_n)
    out["all_cores"] = {"value": 1.0 / (dt_mt * scale_mt), "unit": "proofs/s", "cores": threads, "host_cores": cores,
                        "sample": f"same local compute on {threads} host threads (OpenMP tasks: MSM windows, NTT butterflies, lanes; more threads are slower on this container) at "
                                  f"2^{all_cores_log_n} constraints: {dt_mt:.2f} s" + ("" if all_cores_log_n == log_n_full else f", scaled x{scale_mt:.1f}")}
    return out



# ---------------------------------------------------------------------------------------------------------------
# Plonk / Marlin workloads (BASELINE configs[2] / configs[3]): collaborative-zksnark_amd/polyvm.py drives the library
# ---------------------------------------------------------------------------------------------------------------
def _ec_add(F, p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if F.add(p[1], q[1]) == F.zero:
            return None
        lam = F.mul(F.mul(F.add(F.add(p[0], p[0]), p[0]), p[0]), F.inv(F.add(p[1], p[1])))
    else:
        lam = F.mul(F.sub(q[1], p[1]), F.inv(F.sub(q[0], p[0])))
    x = F.sub(F.sub(F.mul(lam, lam), p[0]), q[0])
    return (x, F.sub(F.mul(lam, F.sub(p[0], x)), p[1]))


def verify_openings(czk, ctx, B, out) -> dict:
    """Every KZG opening against its commitment, on the host, with the KNOWN tau of the synthetic SRS (polyvm.GpuBackend):
    C - [v] G == [tau - x] W in affine big-integer arithmetic.  Marlin's batched opening at beta is checked against the folded
    commitment sum_j ch^j C_j; hiding commitments (Marlin's w, z_a, z_b, g_1) carry their blinding evaluation `random_v`, checked with the known gamma."""
    t0 = time.perf_counter()
    q_rinv = pow(1 << 384, -1, Q_MOD)
    r_rinv = pow(1 << 256, -1, R_MOD)

    def pt(aff, inf):
        if inf:
            return None
        v = [sum(int(aff[6 * k + j]) << (64 * j) for j in range(6)) * q_rinv % Q_MOD for k in range(2)]
        return (v[0], v[1])

    def fr(limbs):
        return sum(int(limbs[j]) << (64 * j) for j in range(4)) * r_rinv % R_MOD
    g = pt(ctx.fixed_base_points(czk.CZK_G1, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0], 0)
    neg = lambda p: None if p is None else (p[0], (-p[1]) % Q_MOD)
    checked = 0

    def check(cmts, opening):
        nonlocal checked
        lanes = opening["value"].shape[0]
        for ln in range(lanes):
            C = cmts(ln)
            v = fr(opening["value"][ln])
            W = pt(opening["proof"][0][ln], opening["proof"][1][ln])
            lhs = _ec_add(_Fq, C, neg(_ec_scalar_mul(_Fq, g, v)))
            if "random_v" in opening:      # a hiding commitment: C - [v] G - [random_v] (gamma G) == [tau - x] W (poly-commit/src/kzg10/mod.rs:296-312)
                lhs = _ec_add(_Fq, lhs, neg(_ec_scalar_mul(_Fq, g, fr(opening["random_v"][ln]) * B.gamma % R_MOD)))
            rhs = _ec_scalar_mul(_Fq, W, (B.tau - opening["point"]) % R_MOD) if W is not None else None
            assert lhs == rhs, "KZG opening does not verify"
            checked += 1
    for k, o in out.items():
        if isinstance(o, dict) and o.get("of"):
            c = out[o["of"] + "_cmt"]
            check(lambda ln: pt(c[0][ln], c[1][ln]), o)
    for key in ("open_beta", "open_gamma"):
        if key not in out:
            continue
        terms = out[key]["terms"]                                   # the opened polynomial = sum coef * (committed polynomial)

        def folded(ln, terms=terms):
            acc = None
            for coef, name in terms:
                cm = out[name + "_cmt"]
                public = cm[0].shape[0] == 1                        # a public polynomial enters a share-wise sum on the lifting lanes only
                lanes_out = out[key]["value"].shape[0]
                if not public or lanes_out == 1 or B.lift[ln]:
                    l2 = 0 if public else ln
                    acc = _ec_add(_Fq, acc, _ec_scalar_mul(_Fq, pt(cm[0][l2], cm[1][l2]), coef))
            return acc
        check(folded, out[key])
    return {"results_checked": True, "results_checked_points": checked, "results_check_s": round(time.perf_counter() - t0, 2),
            "results_check": "every KZG opening verified on the host against its commitment with the synthetic SRS's known tau"}


def _polyiop_party_digests(out, lanes_total: int, parties: int, only=None):
    """One SHA-256 per party over its lanes of every commitment / evaluation / opening (public, single-lane entries are part
    of every party's digest): the party-per-rank layout must reproduce the all-lanes-on-one-GPU layout party by party."""
    per = lanes_total // parties

    def walk(h, x, sel):
        if isinstance(x, dict):
            for k in sorted(x):
                h.update(str(k).encode())
                walk(h, x[k], sel)
        elif isinstance(x, (list, tuple)):
            for v in x:
                walk(h, v, sel)
        elif isinstance(x, np.ndarray):
            h.update(np.ascontiguousarray(sel(x)).tobytes())
        else:
            h.update(repr(x).encode())
    digests = []
    for p in (range(parties) if only is None else [only]):
        h = hashlib.sha256()
        walk(h, out, (lambda a: a) if only is not None else (lambda a, p=p: a[p * per:(p + 1) * per] if a.ndim and a.shape[0] == lanes_total else a))
        digests.append(h.hexdigest())
    return digests


def run_polyiop(args, czk, parallel, ctx, rank, world, n, size_txt):
    import torch
    from czk_amd import polyvm
    plonk = args.workload == "plonk"
    party = args.layout == "party"
    per_party = 1 if plonk else 2                                          # GSZ: one lane per party; SPDZ: sh + mac
    if plonk:
        lanes = per_party if party else args.parties
        lift = None                                                        # public addends on every lane
        max_deg = polyvm.plonk_max_degree(n)
        make_inputs, prove = (lambda B: polyvm.plonk_inputs(B, n)), polyvm.plonk_prove
        if getattr(args, "real_instance", False):
            from polyiop_real import satisfied_plonk_inputs
            make_inputs = lambda B: satisfied_plonk_inputs(B, polyvm, n, 0x51A7 + n)[0]      # noqa: E731
        scheme, what = "GSZ", f"mpc-plonk Prover::prove, {n} gates (wire domain 3 x {size_txt}, mixed radix)"
    else:
        lanes = per_party if party else 2 * args.parties
        king = (1, 1) if rank == 0 else (0, 0)                             # public addends on the king's lanes only
        lift = king if party else tuple([1, 1] + [0] * (2 * args.parties - 2))
        max_deg = polyvm.marlin_max_degree(n)
        make_inputs, prove = (lambda B: polyvm.marlin_inputs(B, n)), polyvm.marlin_prove
        if getattr(args, "real_instance", False) and not party:
            # a real index of a satisfied instance; public data lifted onto EVERY lane, so that each lane is the plain prover and the timed proofs verify
            from polyiop_real import marlin_real_inputs
            lift = tuple([1] * lanes)
            make_inputs = lambda B: marlin_real_inputs(B, polyvm, polyvm.next_pow2(n), 0x3A21 + polyvm.next_pow2(n))      # noqa: E731
        scheme, what = "SPDZ", f"Marlin AHP rounds + commitments + batched openings, {n} constraints"
    t0 = time.time()
    polyvm.GpuBackend.evaluate_by_division = bool(getattr(args, "eval_by_division", False))
    polyvm.GpuBackend.ntt_copy_first = bool(getattr(args, "ntt_copy_first", False))
    B = polyvm.GpuBackend(czk, ctx, lanes, max_deg, lift=lift)
    if party:
        # one party per rank: every batch of evaluations made between two challenges is opened over torch.distributed
        if plonk:
            B.opener = lambda bk, v: parallel.gsz_batch_open(ctx, v[:, 0].contiguous(), degree=(args.parties - 1) // 2)   # t = (n - 1) / 2 (share/gsz20/mod.rs:94-96)
        else:
            mac_share = polyvm.mont(1 if rank == 0 else 0)
            B.opener = lambda bk, v: parallel.spdz_batch_open(ctx, v[:, 0].contiguous(), v[:, 1].contiguous(), mac_share, commit=args.commit_opens)
    B.prepare(polyvm.plonk_commit_sizes(n) if plonk else polyvm.marlin_commit_sizes(n))   # secondary table sets built at SRS load (czk_bases_prepare)
    inp = make_inputs(B)                                            # circuit / index and share lanes: resident in HBM before the timed region
    ctx.sync()
    setup_s = time.time() - t0
    # Independent proofs in flight on this GPU (replica layout): each prover has its own context, stream and share lanes and
    # shares only the registered SRS.  The reference's transcript forces a drain of the MSM pipeline before every challenge; a
    # second proof fills those bubbles -- the same thing the Groth16 bench does by pipelining consecutive proofs.
    want_inflight = args.inflight if args.inflight is not None else 4
    inflight = 1 if party else max(1, min(want_inflight, args.steps))
    provers = [(ctx, B, inp, torch.cuda.current_stream())]
    dev_index = torch.cuda.current_device()
    for _ in range(inflight - 1):
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            c2 = czk.Context(torch.cuda.current_device(), ts.cuda_stream, options=args.ctx_options)
            b2 = polyvm.GpuBackend(czk, c2, lanes, max_deg, lift=lift, share_srs=B)
            i2 = make_inputs(b2)
            c2.sync()
        provers.append((c2, b2, i2, ts))

    def barrier():
        parallel.barrier(torch.cuda.synchronize)

    def run(k_steps_each):
        """k_steps_each[i] proofs on prover i, all provers concurrently (one host thread each); returns the last outputs"""
        import threading
        outs = [None] * len(provers)
        errs = []

        def work(i):
            try:
                c, b, ip, ts = provers[i]
                torch.cuda.set_device(dev_index)           # torch's current device is per thread
                with torch.cuda.stream(ts):
                    for _ in range(k_steps_each[i]):
                        outs[i] = prove(b, ip)
                    c.sync()
            except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
                errs.append(e)
        if len(provers) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(provers))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]
        return outs
    t0 = time.perf_counter()
    prove(B, inp)
    ctx.sync()
    first_ms = (time.perf_counter() - t0) * 1e3
    run([max(0, args.warmup - 1)] + [max(1, args.warmup - 1)] * (inflight - 1))
    for c, b, _, _ in provers:
        b.msm_count = b.ntt_count = b.msm_points = 0
        b.opened = []
        c.profile_reset()
        c.profile_enable(True)
    share = [args.steps // inflight + (1 if i < args.steps % inflight else 0) for i in range(inflight)]
    power = PowerSampler(dev_index)
    barrier()
    power.start()
    t0 = time.perf_counter()
    outs = run(share)
    barrier()
    dt = time.perf_counter() - t0
    power.stop()
    power_local = power.report(args.steps, dt)
    out = outs[0]
    for c, _, _, _ in provers:
        c.profile_enable(False)
    per_rank = per_rank_report(parallel, dt, args.steps, dev_index, power_local)
    dt = parallel.max_over_ranks(dt, device="cuda" if args.backend == "nccl" else "cpu")
    checked = {"results_checked": False} if args.no_result_check else verify_openings(czk, ctx, B, out)
    if getattr(args, "real_instance", False) and not party and not args.no_result_check:
        # the timed proofs themselves through the reference verifiers' equations (tests/polyiop_real.py)
        from polyiop_real import marlin_verify
        if not plonk:
            marlin_verify(polyvm, out)
            checked["timed_proof_verifies"] = True
    for o in outs[1:]:                                  # the other in-flight provers run the same deterministic inputs
        assert _polyiop_party_digests(o, lanes, 1, only=0) == _polyiop_party_digests(out, lanes, 1, only=0), "in-flight provers disagree"
    if party:
        mine = _polyiop_party_digests(out, lanes, 1, only=0)[0]
        got = [None] * world
        torch.distributed.all_gather_object(got, (mine, bool(checked["results_checked"])))
        digests = [g[0] for g in got]
        checked["results_checked"] = all(g[1] for g in got) if not args.no_result_check else False
        opened_batches = len(B.opened) // max(1, args.steps)
    else:
        digests = _polyiop_party_digests(out, lanes, args.parties)
        opened_batches = 0
    digest = hashlib.sha256("".join(digests).encode()).hexdigest()
    proofs = args.steps if party else world * args.steps             # party layout: all ranks work on the same proof
    reads = {k: [c.profile_read(k) for c, _, _, _ in provers] for k in ("ntt_pass", "ntt_mixed", "msm_sort", "msm_accumulate_g1", "msm_reduce")}
    acc_ms, acc_n = sum(r[0] for r in reads["msm_accumulate_g1"]), sum(r[1] for r in reads["msm_accumulate_g1"])
    breakdown = {k: sum(r[0] for r in v) / max(1, args.steps) for k, v in reads.items()}
    busy_ms = busy_union_ms([c for c, _, _, _ in provers], ("msm_accumulate_g1", "msm_accumulate_g2"))
    te = B.bases.arith() == 2
    msm_points, msm_count, ntt_count = (sum(getattr(b, a) for _, b, _, _ in provers) for a in ("msm_points", "msm_count", "ntt_count"))
    pts = msm_points / max(1, args.steps)                         # (point, lane) pairs per proof
    alg_bytes = pts * 32 + (msm_points / max(1, msm_count) * 96) * (msm_count / lanes / max(1, args.steps))   # scalars per lane + bases once per MSM
    achieved = alg_bytes * args.steps / (acc_ms / 1e3) / 1e9 if acc_ms > 0 else 0.0
    where = (f"one party per GPU ({lanes} share lane{'s' if lanes > 1 else ''} each); evaluations opened over torch.distributed "
             f"({'GSZ batch_open' if plonk else 'SPDZ two-round batch_open'}, {opened_batches} batches per proof)") if party else \
            f"{scheme} {args.parties} parties as {lanes} share lanes on one GPU"
    res = {
        "metric": f"collaborative {'Plonk' if plonk else 'Marlin'} proofs/sec (BLS12-377, {size_txt} constraints, {scheme} N={args.parties})",
        "value": proofs / dt, "unit": "proofs/s", "n_gpus": world, "ranks_seen_by_backend": world, "backend": args.backend if world > 1 else None,
        "net": (("czk_net " + args.net_transport) if parallel.get_net() is not None else "torch.distributed") if party else None,
        "steps": args.steps, "warmup": args.warmup, "per_rank": per_rank, **power_local,
        "ms_per_step": dt / args.steps * 1e3, "first_proof_ms": first_ms, "higher_is_better": True, "scaling": "strong" if party else "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic", **checked,
        "config": {"workload": f"{what}; {where}; synthetic circuit / index and SRS, fixed Fiat-Shamir challenges, commitments / evaluations settled at "
                               "every point where the reference's transcript draws a challenge (collaborative-zksnark_amd/polyvm.py)",
                   "constraints": n, "parties": args.parties, "share_lanes": lanes, "layout": args.layout, "results_sha256": digest,
                   "commit_opens": (args.commit_opens if (party and not plonk) else None), "proofs_in_flight": inflight, "inputs": ("a satisfied circuit" if plonk else "a real index of a satisfied instance, public data on every lane") + " (tests/polyiop_real.py)" if getattr(args, "real_instance", False) else "work-shaped stand-ins",
                   "ntt_lanes_per_proof": ntt_count / max(1, args.steps), "msms_per_proof": msm_count / max(1, args.steps),
                   "msm_point_lanes_per_proof": pts},
        "roofline": {"bound": "hbm", "kernel": ("k_accumulate_te (G1 bucket accumulation, twisted Edwards extended coordinates, unsaturated limbs)" if te else
                                                "k_accumulate_u (G1 bucket accumulation, XYZZ, unsaturated limbs)"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": acc_ms / max(1, acc_n), "launches": int(acc_n),
                     "note": "integer-VALU bound; the commitments between two transcript points are enqueued asynchronously (czk_msm_async), so their sort / "
                             "accumulate / reduce stages overlap each other and the NTTs that follow"},
        "accumulate_busy_frac": busy_ms / (dt * 1e3), "accumulate_busy_ms_per_step": busy_ms / max(1, args.steps),
        "stream_elapsed_ms_per_step": {**breakdown, "note": STREAM_ELAPSED_NOTE}, "setup_srs_s": setup_s,
    }
    if rank == 0 and world == 1 and not party and not args.no_result_check and not args.no_verify_report:
        # The timed runs use work-shaped stand-in inputs (random polynomials / random index data), whose proofs cannot verify.  One more proof of the SAME
        # size on a satisfied circuit (Plonk) / a real index of a satisfied instance (Marlin), not timed, through the reference verifiers' equations
        # (tests/polyiop_real.py): every KZG opening with the known tau, and Verifier::verify's four identities / the AHP decision on both sumchecks.
        t0 = time.perf_counter()
        try:
            from polyiop_real import marlin_prove_and_verify, plonk_prove_and_verify
            for c, _, _, _ in provers[1:]:
                c.sync()
            if plonk:
                pts = plonk_prove_and_verify(polyvm, B, lambda o: verify_openings(czk, ctx, B, o), n)
                how = "a satisfied circuit of alternating multiply / add gates; Verifier::verify's public-wire, gate, unit-product and wiring identities (mpc-plonk/src/lib.rs:451-590)"
            else:
                Hm = polyvm.next_pow2(n)
                Bv = polyvm.GpuBackend(czk, ctx, 2, max_deg, lift=(1, 1), share_srs=B)    # public data on both lanes: each lane is the plain prover
                pts = marlin_prove_and_verify(polyvm, Bv, lambda o: verify_openings(czk, ctx, Bv, o), Hm)
                how = ("a real index (the indexer's arithmetisation) of a satisfied instance; the outer sumcheck combination is zero at beta, the inner one at gamma, the batched "
                       "openings open what the verifier folds (marlin/src/ahp/mod.rs:115-260)")
            res["proof_verifies"] = {"proof_verifies": True, "size": n, "kzg_openings_checked": pts, "how": how, "seconds": round(time.perf_counter() - t0, 2)}
        except Exception as e:      # noqa: BLE001 -- reported, never fatal to the measurement
            res["proof_verifies"] = {"proof_verifies": False, "error": repr(e)[-400:], "seconds": round(time.perf_counter() - t0, 2)}
    if rank == 0:
        print(json.dumps(res))
    if parallel.get_net() is not None:
        parallel.get_net().close()      # before its context goes away
        parallel.use_net(None)
    if world > 1:
        torch.distributed.destroy_process_group()


def mac_shortcut_report(czk, device, tstream, n_constraints, args, value, check_results) -> dict:
    """The reference's SPDZ multi_scale_pub_group reads `s.sh.val` for BOTH of its MSMs (mpc-algebra/src/share/spdz.rs:441-442), so
    its mac-lane MSM is its sh-lane MSM again.  A binding at that function may therefore run one MSM per party and return the result
    twice -- bit-identical to the reference.  Reported separately and NEVER as `value`: the headline keeps one MSM per share lane, which
    is what SPDZ with distinct MAC scalars needs (and what the reference spends)."""
    import torch
    from czk_amd.provers import Groth16Local
    try:
        ctx2 = czk.Context(device, tstream.cuda_stream)
        p2 = Groth16Local(czk, ctx2, n_constraints, args.parties, mac_msm_from_sh=True)
        for _ in range(max(1, args.warmup)):
            p2.step()
        p2.all_results.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            p2.step(sync=False)
        ctx2.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = {"proofs_per_s": args.steps / dt, "ms_per_proof": dt / args.steps * 1e3, "fraction_of_value": args.steps / dt / value,
               "note": "one MSM per PARTY (over its sh lane), result used for the sh and the mac group share, as the reference's spdz.rs:440-446 "
                       "computes them; witness map unchanged (all share lanes); never `value`"}
        if not args.no_result_check:
            res["results_checked"] = bool(check_results(czk, ctx2, p2, p2.expand_results(p2.all_results[-1]))["results_checked"])
        del p2
        ctx2.close()
        torch.cuda.empty_cache()
        return res
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[-400:]}


def host_demo_exe() -> str:
    """tools/host_demo.bin, compiled with g++ against the built libczk_hip.so when it is older than its sources (the compiled hosts of
    INTEGRATION.md: tools/host_demo.cpp over include/czk.hpp, tools/groth16_host.hpp, tools/polyvm_host.hpp)"""
    exe = os.path.join(ROOT, "tools", "host_demo.bin")
    pkg = os.path.join(ROOT, "collaborative-zksnark_amd")
    deps = [os.path.join(ROOT, "tools", f) for f in ("host_demo.cpp", "groth16_host.hpp", "polyvm_host.hpp")] + \
           [os.path.join(ROOT, "include", f) for f in ("czk.h", "czk.hpp")] + [os.path.join(pkg, "libczk_hip.so")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "host_demo.cpp"), "-L" + pkg,
                               "-lczk_hip", "-lpthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    return exe


def contexts_in_flight_report(czk, torch, device, args, n_constraints, first, ref_aff, contexts: int, steps: int, value: float) -> dict:
    """`contexts` independent Groth16 pipelines on this GPU at once -- each its own czk context, streams, share lanes and key, one host thread each, `steps`
    pipelined proofs per context -- as a deployment that only wants throughput would run it (the polynomial provers' `--inflight`, for Groth16).  `value` is ONE
    context's pipeline, whose accumulate kernels run one after the other; with several contexts the kernels of different proofs share the GPU (a G2 accumulate
    kernel of one beside G1 kernels, transforms and sorts of the others).  Measured: 0.99 - 1.02 x `value` with 2 - 4 contexts (EXPERIMENTS.md section 14) -- one pipeline
    already fills the GPU.  Never `value`: per-kernel launch times stretch when launches overlap, and the roofline block is defined on one launch at a time.  Every context's last proof must equal the headline's in affine."""
    import threading
    from czk_amd.provers import Groth16Local
    ctx0, p0, ts0 = first
    provers = [first]
    t0 = time.perf_counter()
    for _ in range(contexts - 1):
        ts = torch.cuda.Stream()
        with torch.cuda.stream(ts):
            c = czk.Context(device, ts.cuda_stream, options=args.ctx_options)
            p = Groth16Local(czk, c, n_constraints, args.parties, no_tables=args.no_tables, scheme=args.scheme)
            p.msm_order = p0.msm_order
            p.step()
            p.step()
        provers.append((c, p, ts))
    setup_s = time.perf_counter() - t0
    saved = list(p0.all_results)
    for _, p, _ in provers:
        p.all_results.clear()
    errs = []

    def work(i):
        try:
            c, p, ts = provers[i]
            torch.cuda.set_device(device)           # torch's current device is per thread
            with torch.cuda.stream(ts):
                for _ in range(steps):
                    p.step(sync=False)
                c.sync()
        except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
            errs.append(e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(contexts)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if errs:
        raise errs[0]
    equal = True
    for c, p, _ in provers:
        assert len(p.all_results) == steps
        for k, v in p.all_results[-1].items():
            aff = c.jac_to_affine(czk.CZK_G2 if k == "b_g2" else czk.CZK_G1, v)
            equal = equal and bool(np.array_equal(aff[0], ref_aff[k][0]) and np.array_equal(aff[1], ref_aff[k][1]))
    p0.all_results[:] = saved
    for c, p, _ in provers[1:]:
        p.all_results.clear()
        c.close()
    proofs = contexts * steps
    return {"contexts": contexts, "proofs": proofs, "proofs_per_s": proofs / dt, "ms_per_proof": dt / proofs * 1e3, "vs_value": proofs / dt / value,
            "last_proofs_equal_the_headline's": equal, "setup_s_of_the_extra_contexts": setup_s,
            "note": "independent pipelines on one GPU, one czk context and one host thread each, every context with its own key; never `value` (see DESIGN.md section 5)"}


def seam_device_handles(n_constraints: int, parties: int, steps: int, warmup: int, value: float) -> dict:
    """The same step from a torch-free, Python-free host: tools/host_demo.cpp `bench` (C++ over include/czk.hpp, the mirror of the
    Rust shim) builds the same circuit, key and shares from host vectors, uploads the share lanes ONCE into czk_lanes handles,
    runs `steps` pipelined proofs on the resident lanes and downloads the 20 group elements of each -- what a reference caller
    that follows INTEGRATION.md reaches through the C ABI alone.  Runs as a separate process on the same GPU (this process is
    idle meanwhile); compiled here with g++ against the built libczk_hip.so."""
    try:
        exe = host_demo_exe()
        res = subprocess.run([exe, "bench", "--constraints", str(n_constraints), "--parties", str(parties), "--steps", str(steps), "--warmup", str(max(1, warmup))],
                             capture_output=True, text=True, timeout=900)
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode != 0 or not line:
            return {"error": (res.stdout + res.stderr)[-400:]}
        j = json.loads(line[-1])
        j["fraction_of_value"] = j["proofs_per_s"] / value
        j["note"] = ("host vectors -> czk_lanes_upload once -> constraint evaluation, witness map (Beaver local half, both opens), 5 MSMs x share lanes "
                     "on the resident lanes -> 20 points per proof to the host; no torch, no Python: the reference-side binding's route (never `value`)")
        return j
    except Exception as e:      # noqa: BLE001 -- the report must not take the headline down with it
        return {"error": repr(e)[-400:]}


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks to run, one per GPU (self-launches under torch.distributed.run when needed)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="log2(constraints); BASELINE config = 20")
    ap.add_argument("--constraints", type=int, default=None, help="exact constraint count (overrides --log-n; e.g. 10 = BASELINE configs[0])")
    ap.add_argument("--parties", type=int, default=2)
    ap.add_argument("--cpu-sample-log-n", type=int, default=14)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-result-check", action="store_true")
    ap.add_argument("--no-seam-report", action="store_true")
    ap.add_argument("--real-key", action="store_true", help="groth16, any layout: prove under a REAL key (discrete logs from known toxic waste, tests/groth16_real_key.py) and put the "
                                                           "opened proof through the verification equation (`proof_verifies`; in the party layout the parties' shares are gathered on rank 0) "
                                                           "instead of the discrete-log check of the synthetic key")
    ap.add_argument("--real-instance", action="store_true", help="plonk / marlin: time the prover on a SATISFIED circuit / a real index of a satisfied instance (tests/polyiop_real.py) instead of "
                                                                "the work-shaped stand-ins; marlin: the timed proofs themselves go through the AHP verifier's decision")
    ap.add_argument("--verify-report", action="store_true", help="run the `proof_verifies` leg even as a child of another bench.py (the `other_workloads` entries of configs[0])")
    ap.add_argument("--no-verify-report", action="store_true", help="skip `proof_verifies`: one proof of this configuration under a real key against the Groth16 verification equation")
    ap.add_argument("--workload", choices=("groth16", "plonk", "marlin"), default="groth16",
                    help="groth16 (default; BASELINE metric, SPDZ lanes); plonk: mpc-plonk's prover, GSZ lanes, --log-n = log2(gates) (configs[2]: "
                         "--parties 3 --log-n 18); marlin: AHP rounds + commitments + batched openings, SPDZ lanes (configs[3]: --log-n 20)")
    ap.add_argument("--ntt-copy-first", action="store_true", help="A/B (plonk / marlin): radix-2 transforms as copy + czk_ntt_fr instead of czk_ntt_fr_to")
    ap.add_argument("--eval-by-division", action="store_true", help="A/B (plonk / marlin): evaluations as remainders of czk_poly_div_linear instead of czk_poly_evaluate")
    ap.add_argument("--layout", choices=("replica", "party", "split"), default="replica",
                    help="replica (default, BASELINE configs[1]): every GPU proves independently with all parties' lanes on it; "
                         "party: ONE proof, party p's lanes on rank p (--gpus == --parties), opens all-gathered over RCCL; "
                         "split: ONE proof, all lanes on every rank, every MSM split by base range over the ranks (latency when GPUs outnumber "
                         "parties: 1 / N of the accumulation and of the window tables per GPU, partial results added on rank 0)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only for rigs with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for this rank (default LOCAL_RANK)")
    ap.add_argument("--no-tables", action="store_true", help="groth16: register the proving key with CZK_MEM_NO_TABLES (points only, one bucket set per window): "
                                                             "1/13 of the key memory -- what lets 8 party ranks of the 2^22 configuration share ONE GPU")
    ap.add_argument("--commit-opens", dest="commit_opens", action="store_true", default=True,
                    help="party layout (the default, as in the reference: share/spdz.rs:179): dx_t goes through atomic_broadcast (SHA-256 commit-then-open, channel.rs:50-75)")
    ap.add_argument("--no-commit-opens", dest="commit_opens", action="store_false", help="party layout: skip the commit round of the SPDZ open (a weaker protocol than the reference's; reported as commit_opens: false)")
    ap.add_argument("--inflight", type=int, default=None, help="plonk / marlin, replica layout: independent proofs in flight per GPU, each on its own context "
                                                                 "(default 4 with 24 hardware queues: the transcript points of one proof drain the MSM pipeline, the other proofs fill "
                                                                 "the bubbles -- round 4: plonk 172 / 151 / 137 ms per proof with 1 / 2 / 4 in flight, marlin 226 / 198 / 184)")
    ap.add_argument("--scheme", choices=("spdz", "hbc", "gsz"), default="spdz", help="groth16: spdz (default; sh + mac lane per party), hbc (the reference's honest-but-curious "
                                                                                      "additive sharing: one lane per party) or gsz (honest-majority Shamir sharing: one lane per party, "
                                                                                      "products through the king's degree-reduction open) -- mpc-snarks/src/proof.rs:379-387")
    ap.add_argument("--exchange", choices=("ring", "p2p"), default="ring", help="party layout over RCCL: the opens' share exchange as one ring all-gather or as "
                                                                                 "world - 1 grouped point-to-point copies (parallel.set_exchange)")
    ap.add_argument("--net", choices=("torch", "czk", "czk-ipc"), default="torch", help="party layout: the opens' transport.  torch: torch.distributed collectives issued from "
                                                                               "Python (RCCL through torch, or gloo through the host); czk: the library's own communicator "
                                                                               "(czk_net_*, include/czk.h -- what a compiled host calls): RCCL inside the library when --backend nccl, "
                                                                               "shared memory between the ranks' processes otherwise (czk-ipc: device mailboxes mapped between the "
                                                                               "processes with hipIpc instead of host staging); torch.distributed then only carries the communicator id "
                                                                               "and the timing reduction")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE", help="czk_ctx_set_option on every context before any key is registered "
                                                                                             "(e.g. msm_window_g1=18); repeatable")
    ap.add_argument("--msm-order", default=None, help="groth16, A/B: enqueue order of the four witness-only MSMs, e.g. l,a,b_g1,b_g2 (default l,b_g2,a,b_g1)")
    ap.add_argument("--contexts-report", type=int, default=0, help="groth16, one GPU: also time this many independent pipelines (contexts) on the GPU at once and "
                    "report their aggregate as `contexts_in_flight` (default 0 = skip: measured at 0.99 - 1.02 x `value`, EXPERIMENTS.md section 14)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the `other_workloads` report (configs[2], [3] and the configs[4] size as short child runs)")
    ap.add_argument("--no-multi-gpu-report", action="store_true", help="--gpus N > 1, replica layout: skip the party / split layout children rank 0 runs after the replica line")
    ap.add_argument("--report-budget-s", type=float, default=240.0, help="wall-clock budget of the multi-GPU report; children that would start beyond it are skipped")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group check only: no GPU work (CPU test of --gpus N)")
    args = ap.parse_args()

    if args.layout == "party" and args.gpus != args.parties:
        raise SystemExit(f"--layout party needs one rank per party: --gpus {args.gpus}, --parties {args.parties}")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        if os.environ.get("CZK_BENCH_CHILD"):
            raise SystemExit("bench.py: relaunched child has no WORLD_SIZE")
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks")

    import torch
    import czk_amd as czk
    from czk_amd import parallel
    from czk_amd.provers import Groth16Local
    rank, world, local_rank = parallel.env_rank_world()
    # the driver's SCALE run (`--gpus N`, replica layout, default workload): rank 0 follows the replica line with the party / split layout children
    want_report = (world > 1 and args.layout == "replica" and args.workload == "groth16" and not args.no_multi_gpu_report
                   and not os.environ.get("CZK_BENCH_REPORT_CHILD"))
    if args.dry_run:
        parallel.init("gloo")
        seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        dt = parallel.max_over_ranks(0.001 * (rank + 1))
        if world > 1:
            torch.distributed.destroy_process_group()
        if rank == 0:
            line = {"dry_run": True, "n_gpus": world, "ranks_seen_by_backend": seen, "backend": "gloo", "max_over_ranks_s": dt, "layout": args.layout}
            if want_report:          # the shape of the report the first multi-GPU lease will fill in (tests/test_distributed_cpu.py)
                line["multi_gpu_report"] = multi_gpu_report(world, None, True, args.report_budget_s)
            print(json.dumps(line))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    parallel.init(args.backend)    # RCCL; replica layout: only the timing reduction uses it (units are independent)
    ranks_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    assert ranks_seen == world == args.gpus, (ranks_seen, world, args.gpus)
    party_layout = args.layout == "party"
    split_layout = args.layout == "split" and world > 1
    n_constraints = args.constraints if args.constraints is not None else 1 << args.log_n
    size_txt = f"2^{args.log_n}" if args.constraints is None else str(args.constraints)
    # torch's default stream has handle 0, which the C ABI reads as "make a private stream": use an explicit torch
    # stream so that torch's copies and the library's kernels are ordered on ONE stream.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    ctx_options = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.ctx_option}
    args.ctx_options = ctx_options
    ctx = czk.Context(device, tstream.cuda_stream, options=ctx_options)
    assert tstream.cuda_stream != 0
    parallel.set_exchange(args.exchange)
    if args.net != "torch" and args.layout == "party" and world > 1:
        args.net_transport = "ipc" if args.net == "czk-ipc" else "rccl" if args.backend == "nccl" else "shm"
        parallel.use_net(parallel.make_net(ctx, args.net_transport, device=torch.device("cuda", device) if args.backend == "nccl" else None))
    if args.workload != "groth16":
        return run_polyiop(args, czk, parallel, ctx, rank, world, n_constraints, size_txt)
    rk = real_key_for(n_constraints) if args.real_key else None      # (key, key scalars, (r, s) limbs, r, s)
    if rk is not None:
        args.no_result_check = True       # the discrete-log check knows the synthetic key's logs; a real key's proof goes through the verification equation below
    if party_layout:
        prover = Groth16Local(czk, ctx, n_constraints, args.parties, local_parties=[rank], no_tables=args.no_tables, scheme=args.scheme, key_scalars=rk and rk[1])
        prover.commit_opens = args.commit_opens
    else:
        prover = Groth16Local(czk, ctx, n_constraints, args.parties, no_tables=args.no_tables, scheme=args.scheme,
                              base_split=(rank, world) if split_layout else None, key_scalars=rk and rk[1])

    if args.msm_order:
        prover.msm_order = tuple(args.msm_order.split(","))
        assert sorted(prover.msm_order) == ["a", "b_g1", "b_g2", "l"]

    def barrier():
        parallel.barrier(torch.cuda.synchronize)

    # the very first proof on a fresh context also builds the NTT tables and sizes the workspaces
    t0 = time.perf_counter()
    prover.step()
    first_proof_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(0, args.warmup - 1)):
        prover.step()
    # un-pipelined latency of one proof (enqueue -> results on the host)
    barrier()
    t0 = time.perf_counter()
    prover.step()
    latency_ms = (time.perf_counter() - t0) * 1e3
    prover.all_results.clear()
    ctx.profile_reset()
    ctx.profile_enable(True)
    power = PowerSampler(device)
    barrier()
    power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prover.step(sync=False)     # consecutive proofs pipeline on the context's streams
    ctx.sync()                      # delivers every proof's MSM results to its own host buffers
    if split_layout:                # the K partial sums of every MSM -> rank 0, added there (inside the timed region: it is part of a proof)
        combined = parallel.combine_split_results(ctx, czk, prover.all_results)
    barrier()
    dt = time.perf_counter() - t0
    power.stop()
    power_local = power.report(args.steps, dt)      # this rank's GPU over its own timed loop
    ctx.profile_enable(False)
    assert len(prover.all_results) == args.steps and all(r["h"].any() and r["b_g2"].any() for r in prover.all_results)
    per_rank = per_rank_report(parallel, dt, args.steps, device, power_local)
    dt = parallel.max_over_ranks(dt, device="cuda" if args.backend == "nccl" else "cpu")
    proofs = args.steps if (party_layout or split_layout) else world * args.steps      # party / split layout: all ranks work on the same proof
    if split_layout:
        if rank == 0:
            prover.all_results[:] = combined     # from here on rank 0 holds the proofs' group elements; the other ranks only partial sums
        else:
            args.no_result_check = True

    # every pipelined proof works on the same inputs, so all of them must yield the same group elements (compared in
    # affine: summation order inside buckets is not deterministic, Jacobian triples differ) -- guards the pipelining
    ref_aff = None
    for r in (prover.all_results if (rank == 0 or not split_layout) else []):
        aff = {k: ctx.jac_to_affine(czk.CZK_G2 if k == "b_g2" else czk.CZK_G1, v) for k, v in r.items()}
        if ref_aff is None:
            ref_aff = aff
        else:
            for k in aff:
                assert np.array_equal(aff[k][0], ref_aff[k][0]) and np.array_equal(aff[k][1], ref_aff[k][1]), f"pipelined proofs disagree on {k}"
    # MAC-check vectors of the two opens must be all zero (share/spdz.rs:176-183)
    assert args.scheme != "spdz" or not bool(prover.chk.any().item()), "SPDZ MAC check failed"
    # ... and they must be the RIGHT group elements: host-side check against the bases' known discrete logs (every rank
    # checks its own lanes)
    checked = {"results_checked": False}
    if not args.no_result_check:
        checked = check_results(czk, ctx, prover, prover.all_results[-1])
    # digest of the proof's group elements (affine, key order, party order, sh then mac): equal across layouts
    import hashlib
    mine = b"".join(ref_aff[k][0][ln].tobytes() + bytes([int(ref_aff[k][1][ln])]) for k in ("h", "l", "a", "b_g1", "b_g2")
                    for ln in range(prover.lanes)) if (not party_layout and ref_aff is not None) else (None if party_layout else b"")
    if party_layout:
        per_key = {k: b"".join(ref_aff[k][0][ln].tobytes() + bytes([int(ref_aff[k][1][ln])]) for ln in range(prover.lanes))
                   for k in ("h", "l", "a", "b_g1", "b_g2")}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, per_key)
        mine = b"".join(g[k] for k in ("h", "l", "a", "b_g1", "b_g2") for g in gathered)
    digest = hashlib.sha256(mine).hexdigest()
    real_key_report = None
    if rk is not None:
        # the last proof of the timed region, opened and verified: every rank contributes its local parties' sh-lane shares of Proof{a, b, c} and its lanes'
        # part of the h MSM's exponent; rank 0 adds them up (party layout: gathered over the process group; split layout: rank 0 holds the combined sums)
        from groth16_real_key import R_INV
        key, ks, rs_l, r_int, s_int = rk
        t_v = time.perf_counter()
        have_results = rank == 0 or not split_layout
        shares, h_part = None, 0
        if have_results:
            proof = prover.create_proof({k: v.copy() for k, v in prover.all_results[-1].items()}, rs_l[0], rs_l[1])
            shares = {k: [proof[k][prover.lpp * j] for j in range(len(prover.local))] for k in "abc"}
            h_l = prover.ab.cpu().numpy().view(np.uint64)
            h_part = sum(_dot_mod_r(h_l[prover.lpp * j][:prover.D - 1], ks["h"]) for j in range(len(prover.local))) % R_MOD
        if party_layout:
            got = [None] * world
            torch.distributed.all_gather_object(got, (shares, h_part))
            shares = {k: [sh for g in got for sh in g[0][k]] for k in "abc"}
            h_part = sum(g[1] for g in got) % R_MOD
        if rank == 0:
            opened, ninv = open_proof_shares(czk, ctx, shares, args.scheme, args.parties)
            real_key_report = {**verify_opened_proof(czk, ctx, key, r_int, s_int, opened, h_part * R_INV % R_MOD * ninv % R_MOD),
                               "key": "real: discrete logs from known toxic waste (tests/groth16_real_key.py)", "layout": args.layout,
                               "shares_gathered_from_ranks": world if party_layout else 1, "seconds": round(time.perf_counter() - t_v, 2)}

    acc_ms, acc_n = ctx.profile_read("msm_accumulate_g1")
    acc2_ms, acc2_n = ctx.profile_read("msm_accumulate_g2")
    breakdown = {k: ctx.profile_read(k)[0] / max(1, args.steps) for k in
                 ("ntt_pass", "msm_sort", "msm_accumulate_g1", "msm_accumulate_g2", "msm_reduce")}
    busy_ms = busy_union_ms([ctx], ("msm_accumulate_g1", "msm_accumulate_g2"))
    alg_bytes, launches = prover.g1_accumulate_algorithmic_bytes()
    W = prover.h_query.windows()
    madds = args.steps * prover.g1_mixed_additions_per_step(W)       # G1 mixed additions in the timed region
    achieved = (alg_bytes * args.steps) / (acc_ms / 1e3) / 1e9 if acc_ms > 0 else 0.0
    te = prover.h_query.arith() == 2
    mads_g1 = MADS_PER_MIXED_ADD_TE if te else MADS_PER_MIXED_ADD
    mad_gops = mads_g1 * madds / (acc_ms / 1e3) / 1e9 if acc_ms > 0 else 0.0
    # G2: one launch per step over the b_g2 query (N + 1 points of 192 B, `lanes` scalar vectors of 32 B)
    n_g2 = prover.query_len["b_g2"]
    alg2 = n_g2 * 192 + prover.lanes * n_g2 * 32
    madds2 = args.steps * prover.b_g2_query.windows() * prover.lanes * n_g2
    achieved2 = alg2 * args.steps / (acc2_ms / 1e3) / 1e9 if acc2_ms > 0 else 0.0
    mad2_gops = MADS_PER_MIXED_ADD_G2 * madds2 / (acc2_ms / 1e3) / 1e9 if acc2_ms > 0 else 0.0
    # HBM traffic and the effective clock of the dominant kernels: PMC counters cannot be read from inside this process; the figures
    # are per-launch averages of the same command under `rocprofv3 --pmc` (separate passes), stored with the commit they were taken at
    traffic, traffic_src, pmc, traffic_raw, traffic_cal = None, None, {}, None, None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tf = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tf):
            try:
                pmc = json.load(open(tf))
                traffic_raw = pmc.get("msm_accumulate_g1_bytes_per_launch")
                traffic_src = f"profiles/{name}" + (f" @ {pmc['commit']}" if "commit" in pmc else "")
                if "csrc_sha256" in pmc:
                    traffic_src += "; kernel sources unchanged since" if pmc["csrc_sha256"] == csrc_digest() else \
                                   f"; kernel sources CHANGED since (profile {pmc['csrc_sha256']}, now {csrc_digest()})"
                break
            except Exception:
                pmc = {}
    # FETCH_SIZE under-reports: x 1/2 on wide coalesced streams (MI355X_MICROARCH.md), x 0.666 on this kernel's own access pattern -- random 192-byte
    # table entries of a multi-GB table -- measured with a known byte count (tools/fetch_calib.hip, profiles/r05_fetch_size_calibration.json).
    # traffic = fetch / factor + write (WRITE_SIZE uncalibrated, 3 % of the total)
    try:
        traffic_cal = json.load(open(os.path.join(ROOT, "profiles", "r05_fetch_size_calibration.json")))["k_gather"]["factor"]
    except Exception:      # noqa: BLE001
        traffic_cal = None

    def calibrated(tag):
        f, w = pmc.get(f"msm_accumulate_{tag}_fetch_bytes_per_launch"), pmc.get(f"msm_accumulate_{tag}_write_bytes_per_launch")
        if f is None or w is None:
            return pmc.get(f"msm_accumulate_{tag}_bytes_per_launch")
        return f / traffic_cal + w if traffic_cal else f + w
    traffic = calibrated("g1") if pmc else None
    traffic_note = (f"HBM bytes per launch = FETCH_SIZE / {traffic_cal:.3f} + WRITE_SIZE: the FETCH_SIZE factor is this kernel's own (random 192-byte gathers, "
                    "profiles/r05_fetch_size_calibration.json; the guide's 1/2 holds for wide coalesced streams and is reproduced there); raw counter sum in traffic_raw") \
        if traffic_cal else "raw FETCH_SIZE + WRITE_SIZE (no calibration file)"
    if args.no_tables or n_constraints != 1 << 20 or args.parties != 2 or party_layout:
        traffic, traffic_src, pmc, traffic_raw = None, None, {}, None        # the profile is of the default configuration only
    clk1 = pmc.get("k_accumulate_u_effective_clock_ghz")   # GRBM_GUI_ACTIVE / XCDs / duration under the same command
    clk2 = pmc.get("k_accumulate_u2_effective_clock_ghz")

    def valu_view(gops, per_add, adds, ms, clk, comment):
        v = {"mixed_adds_per_s": adds / (ms / 1e3) if ms > 0 else 0.0, "mad_u64_u32_per_mixed_add": per_add, "mad_u64_u32_gops": gops,
             "mad_u64_u32_peak_gops": MAD_PEAK_MEASURED_GOPS, "frac": gops / MAD_PEAK_MEASURED_GOPS,
             "full_rate_valu_peak_gops": VALU_FULL_RATE_GOPS,
             "peak_note": "MI355X_MICROARCH.md: 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s for a full-rate VALU instruction; v_mad_u64_u32 sustains 33.5 - 35.5 T "
                          "in a register-only loop at a measured 2.42 GHz (tools/bank_bench.hip, profiles/r03_interleave.txt): the peak this kernel is priced against. "
                          "Under this kernel the clock drops to effective_clock_ghz (power management): frac_clock_adjusted prices it at that clock",
             "effective_clock_ghz": clk, "nominal_clock_ghz": NOMINAL_CLOCK_GHZ,
             "frac_clock_adjusted": (gops / (MAD_PEAK_MEASURED_GOPS * clk / NOMINAL_CLOCK_GHZ)) if clk else None,
             "comment": comment}
        return v
    out = {
        # BASELINE.json's metric string for the BASELINE configuration; other sizes / party counts say what they are
        "metric": f"collaborative Groth16 proofs/sec (BLS12-377, {size_txt} constraints, {args.scheme.upper()} N={args.parties})",
        "value": proofs / dt,
        "unit": "proofs/s",
        "n_gpus": world,
        "ranks_seen_by_backend": ranks_seen,
        "backend": args.backend if world > 1 else None,
        "net": (("czk_net " + args.net_transport) if parallel.get_net() is not None else "torch.distributed") if party_layout else None,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "per_rank": per_rank,
        # board power of rank 0's GPU averaged over the timed loop (amdgpu hwmon, 10 ms samples) and the proofs that GPU made per kilojoule
        **power_local,
        "latency_ms_single_proof": latency_ms,
        # the reference's own metric is one proof's wall time (mpc-snarks/src/proof.rs:130-135): the un-pipelined rate
        "proofs_per_s_unpipelined": 1000.0 / latency_ms,
        "first_proof_ms": first_proof_ms,
        # a prover that starts from a proving key in memory and proves ONCE (what the reference's benchmark binary does):
        # czk_bases_register of the five queries + the first proof.  (setup_key_s below also counts generating the synthetic key.)
        "one_shot_s": prover.register_s + prover.reserve_s + first_proof_ms / 1e3,
        "register_key_s": prover.register_s,
        "reserve_s": prover.reserve_s,     # czk_ctx_reserve at key load: NTT tables + MSM workspaces that the first proof would otherwise build
        "higher_is_better": True,
        "scaling": "strong" if (party_layout or split_layout) else "weak",
        # Groth16 SPDZ 2 parties 2^20 on 2x GCP n2-standard-2 (1 core each): 328.957 / 317.213 /
        # 320.422 s per proof (mpc-snarks/analysis/data/weak_1_20.csv:21-23) -> 1 / mean = 0.003104 proofs/s
        "vs_baseline": (proofs / dt) / REF_PROOFS_PER_S if n_constraints == 1 << 20 and args.parties == 2 and args.scheme == "spdz" else None,
        "dtype": "u32",
        "data": "synthetic",
        **checked,
        "config": {"workload": f"Groth16 {args.scheme.upper()} {args.parties} parties, BLS12-377, {size_txt} constraints (squaring circuit), "
                               + ("both parties' share-local NTT+MSM on one GPU" if not party_layout else "one party per GPU") +
                               ": " + prover.describe()
                               + ("; `value` counts ONE MSM PER SHARE LANE -- twice the MSM work of the reference's SPDZ path, whose mac-lane MSM repeats its "
                                  "sh-lane MSM on the same scalars (mpc-algebra/src/share/spdz.rs:441-442); the reference-shaped figure (one MSM per party, "
                                  "result used twice) is reported separately as `spdz_mac_msm_from_sh`" if args.scheme == "spdz" else ""),
                   "constraints": n_constraints, "domain": prover.D, "parties": args.parties, "share_lanes": prover.lanes,
                   "parallelism": (f"ONE proof over {world} GPUs, all share lanes on every rank: the witness map runs in full on each (9 % of a proof, no "
                                   "exchange), every MSM is split by base range (1 / N of the window tables and of the accumulation per GPU), the partial "
                                   "sums are gathered and added on rank 0 (SURVEY.md section 8e: intra-party split for latency)") if split_layout else
                                  (f"{world} independent proofs (one per GPU), no data-path collective; consecutive proofs on a GPU are "
                                   "pipelined (ms_per_step = throughput; latency_ms_single_proof = one proof alone)") if not party_layout else
                                  (f"ONE proof over {world} GPUs, party p's two share lanes on rank p; each of the two opens of the witness map is the "
                                   f"reference's two broadcast rounds (sh lanes, then dx_t = mac_share * value - mac) as all-gathers over {args.backend}, "
                                   "sums and the MAC check on device"),
                   "layout": args.layout, "results_sha256": digest, "window_tables": not args.no_tables,
                   "commit_opens": (bool(prover.commit_opens) if (party_layout and args.scheme == "spdz") else None)},
        "roofline": {"bound": "hbm", "kernel": ("k_accumulate_te (G1 bucket accumulation, twisted Edwards extended coordinates, unsaturated limbs)" if te else
                                                "k_accumulate_u (G1 bucket accumulation, unsaturated limbs)"), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_raw": traffic_raw, "traffic_note": traffic_note if traffic is not None else None,
                     "traffic_source": traffic_src, "avg_launch_ms": acc_ms / max(1, acc_n), "launches": int(acc_n),
                     "algorithmic_bytes_per_launch": alg_bytes / launches,
                     "note": "integer-VALU bound (v_mad_u64_u32), not HBM bound: see DESIGN.md",
                     "valu": valu_view(mad_gops, mads_g1, madds, acc_ms, clk1,
                                       ("twisted Edwards unified mixed addition = 7M, no squarings, no exception handling (csrc/te.h); unsaturated 14x28-bit limbs: "
                                        "7 multiplies x 378 v_mad_u64_u32 = 2646 per mixed addition") if te else
                                       "XYZZ mixed add = 8M+2S; unsaturated 14x28-bit limbs: 6 multiplies x 378 v_mad_u64_u32, 2 squarings x 287, Y3 as two products "
                                       "under one reduction (574) -> 3416 per mixed addition, no carry instructions (csrc/fqu.h)")},
        # the second kernel of the critical stream: the same view for the G2 bucket accumulation (one launch per proof)
        "roofline_g2": {"bound": "hbm", "kernel": "k_accumulate_u2 (G2 bucket accumulation, Fq2 over unsaturated limbs)", "achieved": achieved2, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved2 / HBM_PEAK_GBS, "traffic": calibrated("g2") if pmc else None, "traffic_raw": pmc.get("msm_accumulate_g2_bytes_per_launch"),
                        "traffic_source": traffic_src,
                        "avg_launch_ms": acc2_ms / max(1, acc2_n), "launches": int(acc2_n), "algorithmic_bytes_per_launch": alg2,
                        "valu": valu_view(mad2_gops, MADS_PER_MIXED_ADD_G2, madds2, acc2_ms, clk2,
                                          "Fq2 XYZZ mixed add: 6 Fq2 products (schoolbook, one reduction per component: 1148), 2 Fq2 squarings (756), Y3 as two "
                                          "four-product sums (966 each) -> 10332 v_mad_u64_u32 per mixed addition (csrc/fqu.h)")},
        # time with an accumulate kernel running (union of the HIP-event intervals of the two accumulate kernels) / wall time of the timed region
        "accumulate_busy_frac": busy_ms / (dt * 1e3), "accumulate_busy_ms_per_step": busy_ms / max(1, args.steps),
        "stream_elapsed_ms_per_step": {**breakdown, "note": STREAM_ELAPSED_NOTE},
        "setup_key_s": prover.setup_key_s,
    }
    if (rank == 0 and world == 1 and args.contexts_report > 1 and not party_layout and not split_layout and ref_aff is not None
            and not os.environ.get("CZK_BENCH_CHILD")):
        try:
            out["contexts_in_flight"] = contexts_in_flight_report(czk, torch, device, args, n_constraints, (ctx, prover, tstream), ref_aff, args.contexts_report, args.steps,
                                                                  proofs / dt)
        except Exception as e:      # noqa: BLE001 -- the report must not take the headline down with it
            out["contexts_in_flight"] = {"error": repr(e)[-400:]}
            torch.cuda.empty_cache()
    if real_key_report is not None:
        out["proof_verifies"] = real_key_report
        out["data"] = "synthetic circuit, REAL proving key (known toxic waste)"
    if rank == 0 and world == 1 and not args.no_seam_report and args.scheme == "spdz":
        t = prover.seam_calls_host_memory()
        out["seam_host_memory"] = {"ms_per_proof": t * 1e3, "proofs_per_s": 1.0 / t,
                                   "note": "7 czk_ntt_fr + 5 czk_msm calls per proof with CZK_MEM_HOST (pageable) buffers for all share lanes, "
                                           "bases registered: what a reference caller binding only the NTT / MSM seams sees (PCIe staging included; "
                                           "never `value`)"}
    if args.scheme != "spdz":
        args.no_seam_report = True     # the seam / shortcut reports below are SPDZ-shaped
    if rank == 0 and world == 1 and not args.no_seam_report and not party_layout:
        out["spdz_mac_msm_from_sh"] = mac_shortcut_report(czk, device, tstream, n_constraints, args, proofs / dt, check_results)
    if rank == 0 and world == 1 and not args.no_seam_report and not party_layout:
        out["seam_device_handles"] = seam_device_handles(n_constraints, args.parties, args.steps, args.warmup, proofs / dt)
    if rank == 0 and world == 1 and not args.no_seam_report and not party_layout:
        # the same one-shot figure with the key registered WITHOUT window tables (CZK_MEM_NO_TABLES): a fresh context, so its
        # first proof also builds the NTT tables and sizes the workspaces, like a fresh process would
        del prover
        torch.cuda.empty_cache()
        ctx1 = czk.Context(device, tstream.cuda_stream)
        p1 = Groth16Local(czk, ctx1, n_constraints, args.parties, no_tables=True)
        t0 = time.perf_counter()
        p1.step()
        t1 = time.perf_counter() - t0
        r1 = p1.all_results[-1] if p1.all_results else None
        out["one_shot_no_tables"] = {"seconds": p1.register_s + p1.reserve_s + t1, "register_key_s": p1.register_s, "proof_ms": t1 * 1e3,
                                     "note": "proving key registered with CZK_MEM_NO_TABLES (points only; each MSM runs one bucket set per window): "
                                             "what a prove-once caller should use"}
        if r1 is not None and not args.no_result_check:
            out["one_shot_no_tables"]["results_checked"] = bool(check_results(czk, ctx1, p1, r1)["results_checked"])
    if rank == 0 and world == 1 and not args.no_result_check and not args.no_verify_report and not party_layout and not split_layout and (not os.environ.get("CZK_BENCH_CHILD") or args.verify_report) and rk is None:
        try:
            del prover
        except NameError:
            pass
        try:
            del p1
            ctx1.close()
        except NameError:
            pass
        torch.cuda.empty_cache()
        try:
            out["proof_verifies"] = verify_report(czk, torch, device, tstream, n_constraints, args.parties, args.scheme)
        except Exception as e:      # noqa: BLE001 -- a failed check is reported in the line, it must not take the measurement down with it
            out["proof_verifies"] = {"proof_verifies": False, "error": repr(e)[-400:]}
            torch.cuda.empty_cache()
    if (rank == 0 and world == 1 and not args.no_other_workloads and not party_layout and not split_layout and not args.no_tables and n_constraints == 1 << 20 and args.parties == 2
            and args.scheme == "spdz" and not os.environ.get("CZK_BENCH_CHILD")):
        try:
            del p1
            ctx1.close()
        except NameError:
            pass
        torch.cuda.empty_cache()
        out["other_workloads"] = other_workloads_report(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_n, (n_constraints - 1).bit_length(), args.parties)
    if parallel.get_net() is not None:
        parallel.get_net().close()      # before its context goes away
        parallel.use_net(None)
    if want_report:
        # every rank gives its GPU back (key, tables, workspaces) and leaves the process group; rank 0 alone goes on to run the other
        # layouts as children over the same N GPUs and folds their lines into the one JSON line it prints
        try:
            del prover
        except NameError:
            pass
        ctx.close()
        torch.cuda.empty_cache()
        barrier()
    if world > 1:
        torch.distributed.destroy_process_group()
    if want_report and rank == 0:
        # the replica line FIRST, complete and flushed: a report child that fails or hangs (the RCCL and hipIpc-across-devices legs have never run on
        # hardware) can then cost the report, never the measurement.  The same line follows with the report folded in; a reader takes the last one.
        print(json.dumps({**out, "multi_gpu_report": {"pending": "this line is repeated below with the report folded in"}}), flush=True)
        try:
            out["multi_gpu_report"] = multi_gpu_report(world, out, False, args.report_budget_s)
        except BaseException as e:      # noqa: BLE001 -- incl. KeyboardInterrupt / SystemExit from a child handler: the line below must still be printed
            out["multi_gpu_report"] = {"error": repr(e)[-400:]}
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

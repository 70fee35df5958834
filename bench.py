#!/usr/bin/env python3
"""bench.py -- collaborative Groth16 proofs/sec (BLS12-377, 2^20 constraints, SPDZ, 2 parties) on MI355X.

One "step" = the complete per-party local compute of ONE proof for BOTH parties of a 2-party SPDZ prover on one
GPU (BASELINE.json configs[1]): the R1CS->QAP witness map (7 share-vector NTT ops per party = 28 Fr NTT lanes of
D = 2^21, the Beaver local half and the pointwise steps) plus the 5 MSMs h / l / a / b_g1 / b_g2 for every share
lane (4 lanes: 2 parties x {sh, mac}), inputs already resident in HBM, outputs = the 20 MSM results on the host.
With N GPUs every rank proves its own independent proof (weak scaling, no data-path collective: parties and
proofs are independent units); value = N * steps / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (G1 bucket accumulation), timed live with
HIP events on the stream it runs on; `cpu_baseline` times the CPU checker (oracle/, the C restatement of the
reference algorithms, single thread like the reference's build) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the MSM pipeline uses 3 internal streams (csrc/core.hip)

import numpy as np
import torch

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
REF_PROOFS_PER_S = 3.0 / (328.957 + 317.213 + 320.422)   # reference's published 2^20 SPDZ-2pc timings
MADS_PER_MIXED_ADD = 7 * 378 + 2 * 287 + (2 * 196 + 182)   # 7 multiplies, 2 squarings, 1 fused two-product multiply (fqu.h)
MAD_PEAK_GOPS = 27000.0  # measured v_mad_u64_u32 lane-ops/s on MI355X (tools/microbench.hip), G lane-ops/s


def to_mont_limbs(vals):
    """python ints (canonical) -> (n,4) uint64 Montgomery limbs (a * 2^256 mod r)."""
    R = (1 << 256) % R_MOD
    buf = b"".join(((v * R) % R_MOD).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


class Groth16Local:
    """Device-resident state + the per-step pipeline (mpc-snarks/src/groth/{r1cs_to_qap.rs:47-113, prover.rs:66-178})."""

    def __init__(self, czk, ctx, log_n: int, parties: int, seed: int = 0xC0FFEE, local_parties=None, exchange=None):
        """local_parties: the MPC parties whose share lanes live on this GPU (default: all of them -- BASELINE
        configs[1]); with a strict subset, `exchange` (parallel.all_gather_shares) plays mpc-net's broadcast in the two
        opens of the witness map."""
        from util import rand_fr_canonical
        self.czk, self.ctx = czk, ctx
        self.N = 1 << log_n
        self.P = parties
        self.local = list(range(parties)) if local_parties is None else list(local_parties)
        self.exchange = exchange
        self.lanes = 2 * len(self.local)              # SPDZ: sh + mac per party (share/spdz.rs:50-53)
        self.log_d = (self.N + 2 - 1).bit_length()    # D = next_pow2(N + num_instance) (r1cs_to_qap.rs:63-65)
        self.D = 1 << self.log_d
        N, D, L = self.N, self.D, self.lanes
        dev = torch.device("cuda")

        # ---- synthetic proving key: P_i = [k_i] G  (seed 0xBA5E5) -------------------
        def mk_bases(group, n, sd, inf_first=False):
            k = torch.from_numpy(rand_fr_canonical(0xBA5E5 + sd, n).view(np.int64)).to(dev)
            aw = 12 if group == czk.CZK_G1 else 24
            pts = torch.empty((n, aw), dtype=torch.int64, device=dev)
            ctx.fixed_base_points(group, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
            inf = torch.zeros(n, dtype=torch.uint8, device=dev)
            if inf_first:
                inf[0] = 1   # b_query[1] (the public output has no B entry) is infinity in the real key
            b = ctx.register_bases(group, pts.data_ptr(), inf.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
            del pts, k
            return b
        t0 = time.time()
        self.h_query = mk_bases(czk.CZK_G1, D - 1, 1)             # groth16/src/generator.rs:156-163
        self.l_query = mk_bases(czk.CZK_G1, N, 2)
        self.a_query = mk_bases(czk.CZK_G1, N + 1, 3)             # a_query[1..]
        self.b_g1_query = mk_bases(czk.CZK_G1, N + 1, 4, True)
        self.b_g2_query = mk_bases(czk.CZK_G2, N + 1, 5, True)
        self.setup_key_s = time.time() - t0

        # ---- squaring circuit witness (proof.rs:304-344) and its additive shares ---------------------------
        w = [rand_fr_canonical(seed, 1)[0]]
        w0 = sum(int(w[0][j]) << (64 * j) for j in range(4))
        chain = [w0]
        for _ in range(N):
            chain.append(chain[-1] * chain[-1] % R_MOD)
        wm = to_mont_limbs(chain)                                  # w_0 .. w_N (w_N = public output)
        one = to_mont_limbs([1])[0]
        # additive sharing on the GPU: parties 0..P-2 uniform, last = value - sum (share/spdz.rs:150-162)
        wd = torch.from_numpy(wm.view(np.int64)).to(dev)
        sh = []
        rest = wd.clone()
        for p in range(parties - 1):
            r = torch.from_numpy(rand_fr_canonical(seed + 17 * (p + 1), N + 1).view(np.int64)).to(dev)
            rm = torch.empty_like(r)
            ctx.fr_from_repr(r.data_ptr(), out=rm.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
            ctx.fr_vec_op(1, rest.data_ptr(), rm.data_ptr(), out=rest.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
            sh.append(rm)
        sh.append(rest)
        ctx.sync()
        one_t = torch.from_numpy(one.view(np.int64)).to(dev)

        def lanes_buf():
            return torch.zeros((L, D, 4), dtype=torch.int64, device=dev)
        # a_i = b_i = w_i, c_i = w_{i+1} for i < N; a[N] = 1 (king only: Public lifted per SURVEY a18), a[N+1] = out
        # (a0, b0, c0 are written out directly here as the EXPECTED constraint evaluations: step() computes them on the GPU
        # from `full` with czk_r1cs_matvec; the parity test and the integrity check compare against these)
        self.a0, self.b0, self.c0 = lanes_buf(), lanes_buf(), lanes_buf()
        self.wit = torch.zeros((L, N, 4), dtype=torch.int64, device=dev)          # l-MSM scalars: witness
        self.asg = torch.zeros((L, N + 1, 4), dtype=torch.int64, device=dev)      # a/b-MSM scalars: [out, witness]
        for j, p in enumerate(self.local):
            for m in range(2):                                     # mac lane = sh * mac(), mac() = 1 (spdz.rs:41-47)
                ln = 2 * j + m
                self.a0[ln, :N] = sh[p][:N]
                self.b0[ln, :N] = sh[p][:N]
                self.c0[ln, :N] = sh[p][1:N + 1]
                if p == 0:
                    self.a0[ln, N] = one_t
                self.a0[ln, N + 1] = sh[p][N]
                self.wit[ln] = sh[p][:N]
                self.asg[ln, 0] = sh[p][N]
                self.asg[ln, 1:] = sh[p][:N]
        # full assignment [1, out | w_0 .. w_{N-1}] per lane (r1cs_to_qap.rs:56-61); Public(1) lifted to the king's lanes
        self.full = torch.zeros((L, N + 2, 4), dtype=torch.int64, device=dev)
        for j, p in enumerate(self.local):
            for m in range(2):
                ln = 2 * j + m
                if p == 0:
                    self.full[ln, 0] = one_t
                self.full[ln, 1] = sh[p][N]
                self.full[ln, 2:] = sh[p][:N]
        # the squaring circuit's matrices (proof.rs:304-344): a_i = b_i = w_i, c_i = w_{i+1} (c_{N-1} = out), all coefficients 1
        ones = np.tile(one, (N + 2, 1))
        rp = np.arange(N + 3, dtype=np.uint64)
        wcols = np.arange(2, N + 2, dtype=np.uint32)
        self.mat_a = ctx.r1cs_matrix_register(rp, np.concatenate([wcols, np.array([0, 1], dtype=np.uint32)]), ones, N + 2)
        self.mat_b = ctx.r1cs_matrix_register(rp[: N + 1], wcols, ones[:N], N + 2)
        self.mat_c = ctx.r1cs_matrix_register(rp[: N + 1], np.concatenate([wcols[1:], np.array([1], dtype=np.uint32)]), ones[:N], N + 2)
        # dummy Beaver triples (wire/field.rs:41-60): king holds (1,1,1), everyone else (0,0,0)
        self.tx, self.ty, self.tz = lanes_buf(), lanes_buf(), lanes_buf()
        self.king_lanes = [2 * j + m for j, p in enumerate(self.local) if p == 0 for m in range(2)]
        for t in (self.tx, self.ty, self.tz):
            for ln in self.king_lanes:
                t[ln, :] = one_t
        self.a, self.b, self.c = lanes_buf(), lanes_buf(), lanes_buf()
        self.sx, self.oy = (torch.zeros((D, 4), dtype=torch.int64, device=dev) for _ in range(2))
        self.chk = torch.zeros((2, D, 4), dtype=torch.int64, device=dev)
        self.ab = lanes_buf()
        self.results = {}
        self.all_results = []

    # one open of a share vector: value = sum of sh lanes; MAC check vector = mac_share*value - sum(mac lanes)
    def _open(self, shares, out, chk):
        czk, ctx, D = self.czk, self.ctx, self.D
        ADD, SUB = 0, 1
        M = czk.CZK_MEM_DEVICE
        if len(self.local) < self.P:
            # party-per-GPU layout: mpc-net's broadcast (multi.rs:145-173) is an all-gather of every party's (sh, mac)
            # lanes over RCCL; the sums and the MAC comparison of batch_open (spdz.rs:166-185) are one fused kernel
            assert len(self.local) == 1
            gathered = self.exchange(shares)                       # (P, 2, D, 4), rank order == party order
            bad = ctx.fr_spdz_open(gathered.data_ptr(), self.P, D, out.data_ptr())
            assert bad == 0, "SPDZ MAC check failed"
            return
        ctx.fr_vec_op(ADD, shares[0].data_ptr(), shares[2].data_ptr(), out=out.data_ptr(), n=D, mem=M)
        for p in range(2, self.P):
            ctx.fr_vec_op(ADD, out.data_ptr(), shares[2 * p].data_ptr(), out=out.data_ptr(), n=D, mem=M)
        ctx.fr_vec_op(SUB, out.data_ptr(), shares[1].data_ptr(), out=chk.data_ptr(), n=D, mem=M)
        for p in range(1, self.P):
            ctx.fr_vec_op(SUB, chk.data_ptr(), shares[2 * p + 1].data_ptr(), out=chk.data_ptr(), n=D, mem=M)

    def new_results(self):
        L = self.lanes
        r = {k: np.zeros((L, 18), dtype=np.uint64) for k in ("h", "l", "a", "b_g1")}
        r["b_g2"] = np.zeros((L, 36), dtype=np.uint64)
        return r

    def step(self, sync=True):
        """One proof's local compute.  sync=False only enqueues (consecutive proofs then pipeline: the next proof's
        witness-only MSMs and NTTs overlap this proof's tail); its results are valid after the next ctx.sync()."""
        czk, ctx = self.czk, self.ctx
        D, N, L, ld = self.D, self.N, self.lanes, self.log_d
        M = czk.CZK_MEM_DEVICE
        ADD = 0
        MONT = czk.CZK_SCALAR_MONTGOMERY
        r = self.results = self.new_results()      # every proof keeps its own output buffers
        self.all_results.append(r)
        # --- create_proof MSMs that depend only on the witness (prover.rs:108, 132-156): enqueue-only; they pipeline on
        # the context's internal streams and overlap with the witness map below.  Results are valid after sync().
        ctx.msm_async(self.b_g2_query, self.asg.data_ptr(), N + 1, L, MONT, r["b_g2"], stable=True)
        ctx.msm_async(self.l_query, self.wit.data_ptr(), N, L, MONT, r["l"], stable=True)
        ctx.msm_async(self.a_query, self.asg.data_ptr(), N + 1, L, MONT, r["a"], stable=True)
        ctx.msm_async(self.b_g1_query, self.asg.data_ptr(), N + 1, L, MONT, r["b_g1"], stable=True)
        # --- R1CStoQAP::witness_map ---------------------------------------------------------------------
        # constraint evaluation <A_i, z>, <B_i, z>, <C_i, z> over the share lanes of the full assignment (r1cs_to_qap.rs:
        # 67-83, 95-100); A carries the two instance-copy rows (:79-83).  Rows beyond each matrix are zero padding that the
        # first NTT pass supplies itself, so nothing is cleared or copied.
        ctx.r1cs_matvec(self.mat_a, self.full.data_ptr(), lanes=L, out=self.a.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_b, self.full.data_ptr(), lanes=L, out=self.b.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_c, self.full.data_ptr(), lanes=L, out=self.c.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.witness_map_pre(self.a.data_ptr(), self.b.data_ptr(), ld, L, a_len=N + 2, b_len=N)   # ifft, ifft, coset_fft, coset_fft
        # batch_product_in_place -> S::batch_mul (share/field.rs:97-127): (s + x), (o + y), two opens, combine
        ctx.fr_vec_op(ADD, self.a.data_ptr(), self.tx.data_ptr(), out=self.a.data_ptr(), n=L * D, mem=M)
        ctx.fr_vec_op(ADD, self.b.data_ptr(), self.ty.data_ptr(), out=self.b.data_ptr(), n=L * D, mem=M)
        self._open(self.a, self.sx, self.chk[0])
        self._open(self.b, self.oy, self.chk[1])
        for ln in range(L):
            ctx.fr_beaver_combine(self.tx[ln].data_ptr(), self.ty[ln].data_ptr(), self.tz[ln].data_ptr(), self.sx.data_ptr(),
                                  self.oy.data_ptr(), ln in self.king_lanes, out=self.ab[ln].data_ptr(), n=D, mem=M)
        ctx.witness_map_post(self.ab.data_ptr(), self.c.data_ptr(), ld, L, c_len=N)     # h = ab
        # --- the h MSM (prover.rs:104) needs the witness map's output; NOT flagged stable: the next proof's witness
        # map overwrites `ab`, so the context's stream waits for this MSM's digit extraction (library-side ordering)
        ctx.msm_async(self.h_query, self.ab.data_ptr(), D, L, MONT, r["h"])
        if sync:
            ctx.sync()

    def g1_accumulate_algorithmic_bytes(self):
        """an MSM of n points moves n*(96 B base) once + n*32 B of scalars per lane."""
        tot = 0
        for n in (self.D - 1, self.N, self.N + 1, self.N + 1):
            tot += n * 96 + self.lanes * n * 32
        return tot, 4   # bytes per step, launches per step


def cpu_baseline(log_n_sample: int, log_n_full: int, parties: int):
    """Times the CPU checker (oracle/libczk_oracle.so: limb-exact C restatement of the reference algorithms,
    same Pippenger window rule, same io/oi FFT, single thread like the reference's build) on ONE proof's local
    compute for all share lanes at N = 2^log_n_sample, then scales linearly in N (the reference's own data is
    linear in N: SURVEY.md section 6)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from util import rand_fr_canonical
    N = 1 << log_n_sample
    log_d = (N + 1).bit_length()
    D = 1 << log_d
    lanes = 2 * parties
    # bases: cheap on-curve points for timing: multiples of the generator by small increments via the checker
    from pyref import G1_GEN, G2_GEN, fq_to_mont
    g1 = orc.ints_to_limbs([fq_to_mont(G1_GEN[0]), fq_to_mont(G1_GEN[1])], 6).reshape(-1)
    g2 = orc.ints_to_limbs([fq_to_mont(G2_GEN[0][0]), fq_to_mont(G2_GEN[0][1]), fq_to_mont(G2_GEN[1][0]),
                            fq_to_mont(G2_GEN[1][1])], 6).reshape(-1)

    def chain(g, gen, n):   # P_{i+1} = 2 P_i + G: distinct subgroup points, O(n) group ops
        pts = np.zeros((n, gen.size), dtype=np.uint64)
        jac = np.concatenate([gen, orc.fq_from_repr(orc.ints_to_limbs([1], 6)).reshape(-1) if g == 1 else
                              np.concatenate([orc.fq_from_repr(orc.ints_to_limbs([1], 6)).reshape(-1), np.zeros(6, np.uint64)])])
        for i in range(n):
            aff, _ = orc.jac_to_affine(g, jac)
            pts[i] = aff
            jac = orc.jac_add_mixed(g, orc.jac_double(g, jac), gen)
        return pts
    nb = D
    b1 = chain(1, g1, nb)
    b2 = chain(2, g2, N + 1)
    inf = np.zeros(nb, dtype=np.uint8)
    x = orc.fr_from_repr(rand_fr_canonical(5, D))
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
    scale = _ref_work(log_n_full) / _ref_work(log_n_sample)
    # (ii) the same sample on all host cores: the 4 witness-map -> h-MSM chains and the 16 witness-only MSMs are
    # independent tasks (SURVEY.md section 8d asks for both figures; the reference's published configuration is (i))
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1

    def chain_task():
        a, b = orc.witness_map_pre(x, x, log_d)
        h = orc.witness_map_post(orc.fr_mul(a, b), x, log_d)
        orc.multi_scalar_mul(1, b1[:D - 1], inf, h)
    tasks = [chain_task] * lanes
    for _ in range(lanes):
        tasks += [lambda: orc.multi_scalar_mul(1, b1[:N], inf, x[:N]), lambda: orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1]),
                  lambda: orc.multi_scalar_mul(1, b1[:N + 1], inf, x[:N + 1])]
    tasks = [lambda: orc.multi_scalar_mul(2, b2[:N + 1], inf, x[:N + 1])] * lanes + tasks      # longest tasks first
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(cores, len(tasks))) as ex:
        for f in [ex.submit(t) for t in tasks]:
            f.result()
    dt_mt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": f"oracle C restatement, 1 thread: full local compute of one proof ({lanes} share lanes: witness map + 5 MSMs each) "
                      f"at 2^{log_n_sample} constraints took {dt:.2f} s; scaled x{scale:.1f} to 2^{log_n_full} by the reference algorithm's "
                      f"field-multiplication count (Pippenger windows shrink with N, so this is below the linear x{1 << (log_n_full - log_n_sample)})",
            "all_cores": {"value": 1.0 / (dt_mt * scale), "unit": "proofs/s", "cores": min(cores, len(tasks)), "host_cores": cores,
                          "sample": f"same sample as {len(tasks)} independent tasks (per lane: witness map -> h MSM chain, 4 other MSMs) on a thread "
                                    f"pool: {dt_mt:.2f} s"}}


def _ref_work(log_n: int) -> float:
    """Fq-multiplication-equivalents of the REFERENCE algorithm for one share lane at N = 2^log_n constraints:
    Pippenger with c = ceil(log2 n)*69/100 + 2 (variable_base.rs:21-25): ceil(253/c) windows x (n mixed adds (11 M) +
    2*(2^c - 1) full adds (16 M)); G2 costs 3x; 7 NTTs of D log2(D)/2 Fr multiplications (one Fr mul ~ 0.45 Fq mul)."""
    N = 1 << log_n
    D = 1 << (N + 1).bit_length()

    def msm(n, g2=False):
        lg = (n - 1).bit_length() if n & (n - 1) else n.bit_length() - 1
        c = 3 if n < 32 else lg * 69 // 100 + 2
        w = -(-253 // c)
        return w * (n * 11 + 2 * ((1 << c) - 1) * 16) * (3 if g2 else 1)
    ntt = 7 * D * (D.bit_length() - 1) / 2 * 0.45
    return msm(D - 1) + msm(N) + 2 * msm(N + 1) + msm(N + 1, True) + ntt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="log2(constraints); BASELINE config = 20")
    ap.add_argument("--parties", type=int, default=2)
    ap.add_argument("--cpu-sample-log-n", type=int, default=14)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", choices=("replica", "party"), default="replica",
                    help="replica (default, BASELINE configs[1]): every GPU proves independently with all parties' lanes on it; "
                         "party: ONE proof, party p's lanes on rank p (WORLD_SIZE == --parties), opens all-gathered over RCCL")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only for rigs with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for this rank (default LOCAL_RANK)")
    args = ap.parse_args()

    import czk_amd as czk
    from czk_amd import parallel
    rank, world, local_rank = parallel.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = local_rank if args.device is None else args.device
    torch.cuda.set_device(device)
    parallel.init(args.backend)    # RCCL; replica layout: only the timing reduction uses it (units are independent)
    party_layout = args.layout == "party"
    if party_layout and world != args.parties:
        raise SystemExit(f"--layout party needs one rank per party: WORLD_SIZE={world}, --parties {args.parties}")
    # torch's default stream has handle 0, which the C ABI reads as "make a private stream": use an explicit torch
    # stream so that torch's copies and the library's kernels are ordered on ONE stream.
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    ctx = czk.Context(device, tstream.cuda_stream)
    assert tstream.cuda_stream != 0
    if party_layout:
        prover = Groth16Local(czk, ctx, args.log_n, args.parties, local_parties=[rank], exchange=parallel.all_gather_shares)
    else:
        prover = Groth16Local(czk, ctx, args.log_n, args.parties)

    def barrier():
        parallel.barrier(torch.cuda.synchronize)

    for _ in range(args.warmup):
        prover.step()
    # un-pipelined latency of one proof (enqueue -> results on the host)
    barrier()
    t0 = time.perf_counter()
    prover.step()
    latency_ms = (time.perf_counter() - t0) * 1e3
    prover.all_results.clear()
    ctx.profile_reset()
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prover.step(sync=False)     # consecutive proofs pipeline on the context's streams
    ctx.sync()                      # delivers every proof's MSM results to its own host buffers
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    assert len(prover.all_results) == args.steps and all(r["h"].any() and r["b_g2"].any() for r in prover.all_results)
    dt = parallel.max_over_ranks(dt, device="cuda" if args.backend == "nccl" else "cpu")
    proofs = args.steps if party_layout else world * args.steps      # party layout: all ranks work on the same proof

    # every pipelined proof works on the same inputs, so all of them must yield the same group elements (compared in
    # affine: summation order inside buckets is not deterministic, Jacobian triples differ) -- guards the pipelining
    ref_aff = None
    for r in prover.all_results:
        aff = {k: ctx.jac_to_affine(czk.CZK_G2 if k == "b_g2" else czk.CZK_G1, v) for k, v in r.items()}
        if ref_aff is None:
            ref_aff = aff
        else:
            for k in aff:
                assert np.array_equal(aff[k][0], ref_aff[k][0]) and np.array_equal(aff[k][1], ref_aff[k][1]), f"pipelined proofs disagree on {k}"
    # MAC-check vectors of the two opens must be all zero (share/spdz.rs:176-183)
    assert not bool(prover.chk.any().item()), "SPDZ MAC check failed"
    # digest of the proof's group elements (affine, key order, party order, sh then mac): equal across layouts
    import hashlib
    mine = b"".join(ref_aff[k][0][ln].tobytes() + bytes([int(ref_aff[k][1][ln])]) for k in ("h", "l", "a", "b_g1", "b_g2")
                    for ln in range(prover.lanes)) if not party_layout else None
    if party_layout:
        per_key = {k: b"".join(ref_aff[k][0][ln].tobytes() + bytes([int(ref_aff[k][1][ln])]) for ln in range(prover.lanes))
                   for k in ("h", "l", "a", "b_g1", "b_g2")}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, per_key)
        mine = b"".join(g[k] for k in ("h", "l", "a", "b_g1", "b_g2") for g in gathered)
    digest = hashlib.sha256(mine).hexdigest()

    acc_ms, acc_n = ctx.profile_read("msm_accumulate_g1")
    breakdown = {k: ctx.profile_read(k)[0] / max(1, args.steps) for k in
                 ("ntt_pass", "msm_sort", "msm_accumulate_g1", "msm_accumulate_g2", "msm_reduce")}
    alg_bytes, launches = prover.g1_accumulate_algorithmic_bytes()
    W = 13   # windows at c = 20 (csrc/msm.hip choose_c for n ~ 2^20..2^21)
    madds = args.steps * W * prover.lanes * ((prover.D - 1) + prover.N + 2 * (prover.N + 1))   # G1 mixed additions in the timed region
    achieved = (alg_bytes * args.steps) / (acc_ms / 1e3) / 1e9 if acc_ms > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (see that file)
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("msm_accumulate_g1_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        # BASELINE.json's metric string for the BASELINE configuration; other sizes / party counts say what they are
        "metric": f"collaborative Groth16 proofs/sec (BLS12-377, 2^{args.log_n} constraints, SPDZ N={args.parties})",
        "value": proofs / dt,
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "latency_ms_single_proof": latency_ms,
        "higher_is_better": True,
        "scaling": "weak",
        # Groth16 SPDZ 2 parties 2^20 on 2x GCP n2-standard-2 (1 core each): 328.957 / 317.213 /
        # 320.422 s per proof (mpc-snarks/analysis/data/weak_1_20.csv:21-23) -> 1 / mean = 0.003104 proofs/s
        "vs_baseline": (proofs / dt) / REF_PROOFS_PER_S if args.log_n == 20 and args.parties == 2 else None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"Groth16 SPDZ {args.parties} parties, BLS12-377, 2^{args.log_n} constraints (squaring circuit), "
                               + ("both parties' share-local NTT+MSM on one GPU" if not party_layout else "one party per GPU") +
                               f": {7 * prover.lanes} Fr NTT lanes of 2^{prover.log_d} + 5 MSMs x {prover.lanes} share lanes per GPU",
                   "constraints": 1 << args.log_n, "domain": prover.D, "parties": args.parties, "share_lanes": prover.lanes,
                   "parallelism": (f"{world} independent proofs (one per GPU), no data-path collective; consecutive proofs on a GPU are "
                                   "pipelined (ms_per_step = throughput; latency_ms_single_proof = one proof alone)") if not party_layout else
                                  (f"ONE proof over {world} GPUs, party p's two share lanes on rank p; the two opens of the witness map are "
                                   f"all-gathers over {args.backend} (mpc-net broadcast) followed by the fused sum + MAC-check kernel"),
                   "layout": args.layout, "results_sha256": digest},
        "roofline": {"bound": "hbm", "kernel": "k_accumulate_u (G1 bucket accumulation, unsaturated limbs)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_launch_ms": acc_ms / max(1, acc_n), "launches": int(acc_n),
                     "algorithmic_bytes_per_launch": alg_bytes / launches,
                     "note": "integer-VALU bound (v_mad_u64_u32), not HBM bound: see DESIGN.md",
                     "valu": {"mixed_adds_per_s": madds / (acc_ms / 1e3) if acc_ms > 0 else 0.0,
                              "fq_mul_equiv_per_s": 10 * madds / (acc_ms / 1e3) if acc_ms > 0 else 0.0,
                              "mad_u64_u32_gops": MADS_PER_MIXED_ADD * madds / (acc_ms / 1e3) / 1e9 if acc_ms > 0 else 0.0,
                              "mad_u64_u32_peak_gops": MAD_PEAK_GOPS,
                              "comment": "XYZZ mixed add = 8M+2S = 10 Montgomery multiplies; unsaturated 14x28-bit limbs: 378 v_mad_u64_u32 "
                                         "per multiply, 287 per squaring, one reduction shared by the two products of Y3 -> 3416 per "
                                         "mixed addition, no carry instructions (csrc/fqu.h)"}},
        "breakdown_ms_per_step": breakdown,
        "setup_key_s": prover.setup_key_s,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log_n, args.log_n, args.parties)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

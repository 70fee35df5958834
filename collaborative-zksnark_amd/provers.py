"""Host-side callers of the hot path: the per-party LOCAL compute of the reference's MPC provers, driven through the
C ABI (binding.py).  Mirrors the reference's call sequences, not its protocol logic:

  Groth16Local   R1CStoQAP::witness_map (mpc-snarks/src/groth/r1cs_to_qap.rs:47-113) + the MSM sequence of create_proof
                 (mpc-snarks/src/groth/prover.rs:66-178), Beaver local half (mpc-algebra/src/share/field.rs:97-127);
                 create_proof's O(1) group steps and the Proof{a, b, c} share of every lane for public r, s (prover.rs:110-178, 216-232)

Used by bench.py (timed) and tests/ (parity against the checker).  torch provides device memory and streams only.
"""
from __future__ import annotations

import time

import numpy as np
import torch

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
_MASK64 = (1 << 64) - 1


def splitmix_u64(seed: int, n: int) -> np.ndarray:
    """n SplitMix64 outputs (deterministic inputs, independent of rand 0.7)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_R_LIMBS = np.array([(R_MOD >> (64 * i)) & _MASK64 for i in range(4)], dtype=np.uint64)


def rand_fr_canonical(seed: int, n: int) -> np.ndarray:
    """(n,4) uint64 canonical values < r: top 3 bits masked (REPR_SHAVE_BITS, fr.rs:44), candidates >= r rejected
    (fields/arithmetic.rs:199-214).  Same stream as tests/util.py."""
    out = np.zeros((0, 4), dtype=np.uint64)
    chunk = 0
    while out.shape[0] < n:
        m = max(16, int((n - out.shape[0]) * 1.7) + 8)
        raw = splitmix_u64(seed + 0x1000003 * chunk, 4 * m).reshape(m, 4)
        raw[:, 3] &= np.uint64(_MASK64 >> 3)
        lt = np.zeros(m, dtype=bool)
        eq = np.ones(m, dtype=bool)
        for j in (3, 2, 1, 0):
            lt |= eq & (raw[:, j] < _R_LIMBS[j])
            eq &= raw[:, j] == _R_LIMBS[j]
        out = np.vstack([out, raw[lt]])
        chunk += 1
    return np.ascontiguousarray(out[:n])


def to_mont_limbs(vals):
    """python ints (canonical) -> (n,4) uint64 Montgomery limbs (a * 2^256 mod r)."""
    R = (1 << 256) % R_MOD
    buf = b"".join(((v * R) % R_MOD).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177


def to_mont_limbs_fq(v: int) -> np.ndarray:
    """canonical integer -> (6,) uint64 Montgomery limbs of Fq (R = 2^384)"""
    x = v % Q_MOD * ((1 << 384) % Q_MOD) % Q_MOD
    return np.array([(x >> (64 * i)) & _MASK64 for i in range(6)], dtype=np.uint64)


BASE_SEED = 0xBA5E5   # P_i = [k_i] G with k_i = rand_fr_canonical(BASE_SEED + query id, n)  
QUERIES = (("h", 1), ("l", 2), ("a", 3), ("b_g1", 4), ("b_g2", 5))


class Groth16Local:
    """Device-resident state + the per-proof pipeline for `n_constraints` constraints of the reference's benchmark
    circuit (squaring chain, mpc-snarks/src/proof.rs:304-344), SPDZ shares of `parties` parties."""

    def __init__(self, czk, ctx, n_constraints: int, parties: int, seed: int = 0xC0FFEE, local_parties=None, exchange=None, no_tables: bool = False,
                 mac_msm_from_sh: bool = False, scheme: str = "spdz", base_split=None, key_scalars=None):
        """local_parties: the MPC parties whose share lanes live on this GPU (default: all of them -- BASELINE
        configs[1]); with one party per rank the two opens of the witness map run the reference's two broadcast rounds
        over torch.distributed (parallel.spdz_batch_open).  `exchange` is kept for callers that pass it; unused.  no_tables:
        register the proving key with CZK_MEM_NO_TABLES (what a prover that runs once should do).  key_scalars: discrete logs of a proving key to use
        in place of the synthetic one -- {"h": (D - 1, 4), "l": (N, 4), "a" / "b_g1" / "b_g2": (N + 1, 4) [the queries from index 1 on], "pk_g1": (4, 4)
        [alpha, beta, delta, a_query[0]], "pk_g2": (2, 4) [beta, delta]}, canonical uint64 limbs: how tests/test_verify.py hands over a key generated
        from known toxic waste (groth16/src/generator.rs:60-230), so that the proof can be checked against the verification equation."""
        self.czk, self.ctx = czk, ctx
        ks = key_scalars or {}
        # base_split = (k, K): the intra-party split for latency when GPUs outnumber parties ("MSM by base range -> one extra
        # point-add").  This rank registers and sums only bases [n k / K, n (k + 1) / K) of every query -- 1 / K of the window tables and of the
        # accumulation -- and runs the (cheap: 9 % of a proof) witness map in full, so no exchange is needed inside a proof; the K partial results of
        # each MSM are gathered and added on rank 0 (parallel.combine_split_results).  A range's results are partial sums, not proof elements.
        self.base_split = None if base_split is None else (int(base_split[0]), int(base_split[1]))
        # scheme "spdz": two lanes per party (sh, mac; share/spdz.rs:50-53), opens carry the MAC check.  scheme "hbc": the reference's
        # honest-but-curious additive sharing (mpc-snarks/src/proof.rs:379-387 `--alg hbc`; AdditiveFieldShare, share/add.rs:26-29):
        # ONE lane per party, an open is the sum of the parties' lanes (add.rs:256-259), no MAC lane and no check.
        # scheme "gsz": the reference's honest-majority Shamir sharing (`--alg gsz`; GszFieldShare, share/gsz20/mod.rs:115-118): ONE lane per party
        # holding p(w^j) of a degree-t polynomial with p(0) = the value (t = (n - 1) / 2, :94-96); add / scale are lane-wise, a public addend goes to
        # EVERY lane (:266-269 shift), and a product is batch_mult (:556-595): lane-wise x * y + r2, the king opens the degree-2t result and
        # hands the value back to everyone (batch_king_compute with f = identity, :494-527), minus r -- with the reference's stubbed double share
        # r = r2 = 1 (:394-407).  No Beaver triples.
        assert scheme in ("spdz", "hbc", "gsz")
        self.scheme = scheme
        self.gsz_t = (parties - 1) // 2
        self.lpp = 2 if scheme == "spdz" else 1       # share lanes per party
        assert not (mac_msm_from_sh and scheme != "spdz")
        # mac_msm_from_sh: the reference's SPDZ multi_scale_pub_group computes BOTH group shares from the `sh` scalars
        # (mpc-algebra/src/share/spdz.rs:440-446: `macs` is built from `s.sh.val` too), so its mac-lane MSM repeats its sh-lane MSM
        # bit for bit.  A caller that binds at that function can run ONE MSM per party and use the result twice; this switch does
        # exactly that (MSMs over the parties' sh lanes only, results duplicated).  Off by default: the headline keeps one MSM per
        # share lane, as a general SPDZ implementation with distinct MAC scalars needs.
        self.mac_msm_from_sh = bool(mac_msm_from_sh)
        self.N = int(n_constraints)
        self.P = parties
        self.local = list(range(parties)) if local_parties is None else list(local_parties)
        self.exchange = exchange
        # enqueue order of the four MSMs that need no witness map.  The witness map only makes progress beside the G2 accumulate kernel (EXPERIMENTS.md
        # section 14); with that kernel second it is done closer to the moment `h` is needed: 14.51 / 14.54 against 14.42 / 14.29 proofs/s with b_g2 first
        self.msm_order = ("l", "b_g2", "a", "b_g1")
        self.commit_opens = True                      # dx_t goes through atomic_broadcast (commit-then-open, spdz.rs:179, channel.rs:50-75); False = explicit opt-out
        # mac_share() = 1 on the king, 0 elsewhere (share/spdz.rs:30-37: the reference's stand-in MAC key is 1)
        self.mac_share = to_mont_limbs([1 if (self.local and self.local[0] == 0) else 0])[0]
        self.lanes = self.lpp * len(self.local)       # SPDZ: sh + mac per party (share/spdz.rs:50-53); HBC: the additive share alone
        self.log_d = (self.N + 2 - 1).bit_length()    # D = next_pow2(N + num_instance) (r1cs_to_qap.rs:63-65)
        self.D = 1 << self.log_d
        N, D, L = self.N, self.D, self.lanes
        dev = torch.device("cuda")

        # ---- synthetic proving key: P_i = [k_i] G ----------------------------------------------------------
        def mk_bases(group, n, sd, inf_first=False):
            lo, hi = self.base_range(n)
            name = QUERIES[sd - 1][0]
            kk = np.ascontiguousarray(ks[name], dtype=np.uint64).reshape(n, 4) if name in ks else rand_fr_canonical(BASE_SEED + sd, n)
            k = torch.from_numpy(kk[lo:hi].copy().view(np.int64)).to(dev)   # the same bases in every layout
            n = hi - lo
            aw = 12 if group == czk.CZK_G1 else 24
            pts = torch.empty((n, aw), dtype=torch.int64, device=dev)
            ctx.fixed_base_points(group, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
            inf = torch.zeros(n, dtype=torch.uint8, device=dev)
            if inf_first and lo == 0:
                inf[0] = 1   # b_query[1] (the public output has no B entry) is infinity in the real key
            ctx.sync()
            t_reg = time.time()
            b = ctx.register_bases(group, pts.data_ptr(), inf.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE | (czk.CZK_MEM_NO_TABLES if no_tables else 0))
            self.register_s += time.time() - t_reg     # czk_bases_register returns when the tables are built
            del pts, k
            return b
        self.register_s = 0.0                           # czk_bases_register alone; setup_key_s also counts the synthetic key generation
        t0 = time.time()
        self.query_len = {"h": D - 1, "l": N, "a": N + 1, "b_g1": N + 1, "b_g2": N + 1}   # groth16/src/generator.rs:156-163
        self.h_query = mk_bases(czk.CZK_G1, D - 1, 1)
        self.l_query = mk_bases(czk.CZK_G1, N, 2)
        self.a_query = mk_bases(czk.CZK_G1, N + 1, 3)             # a_query[1..]
        self.b_g1_query = mk_bases(czk.CZK_G1, N + 1, 4, True)
        self.b_g2_query = mk_bases(czk.CZK_G2, N + 1, 5, True)
        # the rest of the proving key (groth16/src/data_structures.rs:132-149): vk.alpha_g1, beta_g1, delta_g1, a_query[0] in G1 and
        # vk.beta_g2, vk.delta_g2 in G2, synthetic like the queries; b_g1_query[0] / b_g2_query[0] are infinity in the real key (the constant-one
        # variable has no B entry).  Host values: create_proof uses them in O(1) group steps.
        g1x = ctx.fixed_base_points(czk.CZK_G1, np.ascontiguousarray(ks["pk_g1"], dtype=np.uint64).reshape(4, 4) if "pk_g1" in ks else rand_fr_canonical(BASE_SEED + 6, 4))
        g2x = ctx.fixed_base_points(czk.CZK_G2, np.ascontiguousarray(ks["pk_g2"], dtype=np.uint64).reshape(2, 4) if "pk_g2" in ks else rand_fr_canonical(BASE_SEED + 7, 2))
        self.pk = {"alpha_g1": g1x[0], "beta_g1": g1x[1], "delta_g1": g1x[2], "a_query0": g1x[3], "beta_g2": g2x[0], "delta_g2": g2x[1]}
        self.setup_key_s = time.time() - t0
        # czk_ctx_reserve: NTT tables of the witness-map domain and the MSM workspaces for this key's call shapes, built here (like the window
        # tables above) instead of inside the first proof; reported separately (bench.py: reserve_s, part of one_shot_s)
        t_res = time.time()
        ctx.reserve(self.log_d, L, self.h_query, D, L)
        ctx.reserve(0, 0, self.b_g2_query, N + 1, L)
        ctx.sync()
        self.reserve_s = time.time() - t_res

        # ---- squaring circuit witness (proof.rs:304-344) and its additive shares ---------------------------
        w = [rand_fr_canonical(seed, 1)[0]]
        w0 = sum(int(w[0][j]) << (64 * j) for j in range(4))
        chain = [w0]
        for _ in range(N):
            chain.append(chain[-1] * chain[-1] % R_MOD)
        wm = to_mont_limbs(chain)                                  # w_0 .. w_N (w_N = public output)
        one = to_mont_limbs([1])[0]
        wd = torch.from_numpy(wm.view(np.int64)).to(dev)
        sh = []
        if scheme == "gsz":
            # Shamir sharing on the GPU: party j holds w + sum_{k=1..t} w_n^(j k) c_k, w_n = the order-n root of the share domain (gsz20/mod.rs:98-105),
            # c_k uniform (poly_share, :214-232: a random polynomial of degree t with the value as its constant coefficient)
            wn = ctx.share_domain_constants(parties)["group_gen"]
            wn_int = sum(int(wn[i]) << (64 * i) for i in range(4)) * pow(1 << 256, -1, R_MOD) % R_MOD
            coef = []
            for k in range(1, self.gsz_t + 1):
                c = torch.from_numpy(rand_fr_canonical(seed + 31 * k, N + 1).view(np.int64)).to(dev)
                ctx.fr_from_repr(c.data_ptr(), out=c.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
                coef.append(c)
            for j in range(parties):
                acc = wd.clone()
                for k, c in enumerate(coef, start=1):
                    sc = torch.from_numpy(to_mont_limbs([pow(wn_int, j * k, R_MOD)]).view(np.int64)).to(dev)
                    tmp = torch.empty_like(c)
                    ctx.fr_vec_scale(c.data_ptr(), sc.data_ptr(), out=tmp.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
                    ctx.fr_vec_op(0, acc.data_ptr(), tmp.data_ptr(), out=acc.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
                    ctx.sync()                         # `sc` / `tmp` are read in stream order: keep them alive until the kernels have run
                sh.append(acc)
        else:
            # additive sharing on the GPU: parties 0..P-2 uniform, last = value - sum (share/spdz.rs:150-162)
            rest = wd.clone()
            for p in range(parties - 1):
                r = torch.from_numpy(rand_fr_canonical(seed + 17 * (p + 1), N + 1).view(np.int64)).to(dev)
                rm = torch.empty_like(r)
                ctx.fr_from_repr(r.data_ptr(), out=rm.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
                ctx.fr_vec_op(1, rest.data_ptr(), rm.data_ptr(), out=rest.data_ptr(), n=N + 1, mem=czk.CZK_MEM_DEVICE)
                sh.append(rm)
            sh.append(rest)
        ctx.sync()
        # lanes that take PUBLIC addends (Public + Shared = shift): the king's for additive sharings (spdz.rs:204-208, add.rs:141-146), every lane for
        # Shamir shares (a constant is added to the polynomial: gsz20/mod.rs:266-269)
        pub_party = (lambda p: True) if scheme == "gsz" else (lambda p: p == 0)
        one_t = torch.from_numpy(one.view(np.int64)).to(dev)

        def lanes_buf():
            return torch.zeros((L, D, 4), dtype=torch.int64, device=dev)
        # a_i = b_i = w_i, c_i = w_{i+1} for i < N; a[N] = 1 (king only: Public lifted per SURVEY a18), a[N+1] = out
        # (a0, b0, c0 are written out directly here as the EXPECTED constraint evaluations: step() computes them on the GPU
        # from `full` with czk_r1cs_matvec; the parity test and the integrity check compare against these)
        self.a0, self.b0, self.c0 = lanes_buf(), lanes_buf(), lanes_buf()
        self.wit = torch.zeros((L, N, 4), dtype=torch.int64, device=dev)          # l-MSM scalars: witness
        self.asg = torch.zeros((L, N + 1, 4), dtype=torch.int64, device=dev)      # a/b-MSM scalars: [out, witness]
        lpp = self.lpp
        for j, p in enumerate(self.local):
            for m in range(lpp):                                   # mac lane = sh * mac(), mac() = 1 (spdz.rs:41-47)
                ln = lpp * j + m
                self.a0[ln, :N] = sh[p][:N]
                self.b0[ln, :N] = sh[p][:N]
                self.c0[ln, :N] = sh[p][1:N + 1]
                if pub_party(p):
                    self.a0[ln, N] = one_t
                self.a0[ln, N + 1] = sh[p][N]
                self.wit[ln] = sh[p][:N]
                self.asg[ln, 0] = sh[p][N]
                self.asg[ln, 1:] = sh[p][:N]
        # full assignment [1, out | w_0 .. w_{N-1}] per lane (r1cs_to_qap.rs:56-61); Public(1) lifted to the king's lanes
        self.full = torch.zeros((L, N + 2, 4), dtype=torch.int64, device=dev)
        for j, p in enumerate(self.local):
            for m in range(lpp):
                ln = lpp * j + m
                if pub_party(p):
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
        self.king_lanes = [lpp * j + m for j, p in enumerate(self.local) if pub_party(p) for m in range(lpp)]
        if scheme == "gsz":
            self.ones = torch.zeros((D, 4), dtype=torch.int64, device=dev)      # the stubbed double share r = r2 = 1 on every party (gsz20/mod.rs:394-407)
            self.ones[:] = one_t
        else:
            for t in (self.tx, self.ty, self.tz):
                for ln in self.king_lanes:
                    t[ln, :] = one_t
        self.a, self.b, self.c = lanes_buf(), lanes_buf(), lanes_buf()
        self.sx, self.oy = (torch.zeros((D, 4), dtype=torch.int64, device=dev) for _ in range(2))
        self.chk = torch.zeros((2, D, 4), dtype=torch.int64, device=dev)
        self.ab = lanes_buf()
        if self.mac_msm_from_sh:
            self.wit_sh, self.asg_sh = self.wit[0::2].contiguous(), self.asg[0::2].contiguous()
            self.ab_sh = torch.zeros((L // 2, D, 4), dtype=torch.int64, device=dev)
        self.wit_q, self.asg_q = self._sl(self.wit, N), self._sl(self.asg, N + 1)     # this rank's share of the witness-only MSMs' scalars
        self.results = {}
        self.all_results = []

    def base_range(self, n: int):
        """[lo, hi) of a query of n bases this rank sums (everything without base_split)"""
        if self.base_split is None:
            return 0, n
        k, K = self.base_split
        return n * k // K, n * (k + 1) // K

    def _sl(self, t, n):
        """the scalars of this rank's base range, contiguous per lane (the MSM entry points take lanes x n_scalars without a stride)"""
        lo, hi = self.base_range(n)
        return t if (lo, hi) == (0, n) and t.shape[1] == n and t.is_contiguous() else t[:, lo:hi].contiguous()

    def ntt_lanes_per_step(self):
        return 7 * self.lanes

    def describe(self):
        return (f"{7 * self.lanes} Fr NTT lanes of 2^{self.log_d} + 5 MSMs x {self.lanes} share lanes per GPU"
                + {"spdz": "", "hbc": " (HBC: one additive-share lane per party, no MAC lane)",
                   "gsz": " (GSZ: one Shamir-share lane per party; products by batch_mult with the king's degree-2t open)"}[self.scheme])

    # one open of a share vector: value = sum of sh lanes; MAC check vector = mac_share*value - sum(mac lanes)
    def _open(self, shares, out, chk):
        czk, ctx, D = self.czk, self.ctx, self.D
        ADD, SUB = 0, 1
        M = czk.CZK_MEM_DEVICE
        if len(self.local) < self.P:
            # party-per-GPU layout: the reference's batch_open round by round (share/spdz.rs:166-185) -- broadcast of the `sh`
            # lanes (mpc-net/src/multi.rs:145-173 = one all-gather over RCCL), dx_t = mac_share * value - mac, broadcast of
            # dx_t (atomic_broadcast when self.commit_opens), sum == 0.  MAC shares stay on their party.
            assert len(self.local) == 1
            from . import parallel
            if self.scheme == "hbc":
                vals = parallel.additive_batch_open(ctx, shares[0])
            else:
                vals = parallel.spdz_batch_open(ctx, shares[0], shares[1], self.mac_share, commit=self.commit_opens)
            out.copy_(vals)
            return
        if self.scheme == "hbc" and self.P == 1:     # a single prover: the "share" is the value
            out.copy_(shares[0])
            return
        if self.scheme == "hbc":                     # AdditiveFieldShare::batch_open (share/add.rs:256-259): the sum of the parties' lanes
            ctx.fr_vec_op(ADD, shares[0].data_ptr(), shares[1].data_ptr(), out=out.data_ptr(), n=D, mem=M)
            for p in range(2, self.P):
                ctx.fr_vec_op(ADD, out.data_ptr(), shares[p].data_ptr(), out=out.data_ptr(), n=D, mem=M)
            return
        ctx.fr_vec_op(ADD, shares[0].data_ptr(), shares[2].data_ptr(), out=out.data_ptr(), n=D, mem=M)
        for p in range(2, self.P):
            ctx.fr_vec_op(ADD, out.data_ptr(), shares[2 * p].data_ptr(), out=out.data_ptr(), n=D, mem=M)
        ctx.fr_vec_op(SUB, out.data_ptr(), shares[1].data_ptr(), out=chk.data_ptr(), n=D, mem=M)
        for p in range(1, self.P):
            ctx.fr_vec_op(SUB, chk.data_ptr(), shares[2 * p + 1].data_ptr(), out=chk.data_ptr(), n=D, mem=M)

    def _gsz_batch_mult(self):
        """gsz20::batch_mult (share/gsz20/mod.rs:556-595) on the lanes of a, b -> ab: x.val *= y.val; x.degree *= 2; x.val += r2.val; the king opens
        the degree-2t shares and hands the value to every party (batch_king_compute, f = identity: :494-527); shift_res.val -= r.val."""
        czk, ctx, D, L = self.czk, self.ctx, self.D, self.lanes
        M, ADD, SUB, MUL = czk.CZK_MEM_DEVICE, 0, 1, 2
        ctx.fr_vec_op(MUL, self.a.data_ptr(), self.b.data_ptr(), out=self.ab.data_ptr(), n=L * D, mem=M)
        for ln in range(L):
            ctx.fr_vec_op(ADD, self.ab[ln].data_ptr(), self.ones.data_ptr(), out=self.ab[ln].data_ptr(), n=D, mem=M)
        if len(self.local) < self.P:                 # one party per rank: send_to_king, open on the king, recv_from_king
            assert len(self.local) == 1
            from . import parallel
            self.sx.copy_(parallel.gsz_batch_king_compute(ctx, self.ab[0], 2 * self.gsz_t))
        else:                                        # every party's lane is on this GPU: the king's open is local, its answer is every lane's new share
            bad = ctx.fr_gsz_open(self.ab.data_ptr(), self.P, D, self.sx.data_ptr(), degree=2 * self.gsz_t)
            if bad:
                raise RuntimeError(f"GSZ degree check failed on {bad} of {D} products (assert!(p.degree() <= d), share/gsz20/mod.rs:452)")
        for ln in range(L):
            ctx.fr_vec_op(SUB, self.sx.data_ptr(), self.ones.data_ptr(), out=self.ab[ln].data_ptr(), n=D, mem=M)

    def new_results(self):
        L = self.lanes // 2 if self.mac_msm_from_sh else self.lanes
        r = {k: np.zeros((L, 18), dtype=np.uint64) for k in ("h", "l", "a", "b_g1")}
        r["b_g2"] = np.zeros((L, 36), dtype=np.uint64)
        return r

    def expand_results(self, r):
        """mac_msm_from_sh: one result per party -> the (sh, mac) pair of every party, as the reference's multi_scale_pub_group returns"""
        return {k: np.repeat(v, 2, axis=0) for k, v in r.items()} if self.mac_msm_from_sh and r["h"].shape[0] != self.lanes else r

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
        if self.mac_msm_from_sh:
            return self._step_mac_from_sh(r, sync)
        # --- create_proof MSMs that depend only on the witness (prover.rs:108, 132-156): enqueue-only; they pipeline on
        # the context's internal streams and overlap with the witness map below.  Results are valid after sync().
        na, nw = self.asg_q.shape[1], self.wit_q.shape[1]          # N + 1 and N, or this rank's base range of them (base_split)
        early = {"b_g2": lambda s: ctx.msm_async(self.b_g2_query, self.asg_q.data_ptr(), na, L, MONT, r["b_g2"], stable=True, same_scalars=s),
                 "l": lambda s: ctx.msm_async(self.l_query, self.wit_q.data_ptr(), nw, L, MONT, r["l"], stable=True),
                 "a": lambda s: ctx.msm_async(self.a_query, self.asg_q.data_ptr(), na, L, MONT, r["a"], stable=True, same_scalars=s),
                 "b_g1": lambda s: ctx.msm_async(self.b_g1_query, self.asg_q.data_ptr(), na, L, MONT, r["b_g1"], stable=True, same_scalars=s)}
        seen_asg = False
        for k in self.msm_order:      # the accumulate kernels run in this order, `h` after them (EXPERIMENTS.md section 14)
            # a, b_g1 and b_g2 all take `assignment` (prover.rs:130-166): the second and third say so, and the library keeps one digit sort per proof where
            # the keys allow it (CZK_MEM_SAME_SCALARS; the first of the three always sorts)
            early[k](seen_asg and k != "l")
            seen_asg = seen_asg or k != "l"
        # --- R1CStoQAP::witness_map ---------------------------------------------------------------------
        # constraint evaluation <A_i, z>, <B_i, z>, <C_i, z> over the share lanes of the full assignment (r1cs_to_qap.rs:
        # 67-83, 95-100); A carries the two instance-copy rows (:79-83).  Rows beyond each matrix are zero padding that the
        # first NTT pass supplies itself, so nothing is cleared or copied.
        ctx.r1cs_matvec(self.mat_a, self.full.data_ptr(), lanes=L, out=self.a.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_b, self.full.data_ptr(), lanes=L, out=self.b.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_c, self.full.data_ptr(), lanes=L, out=self.c.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.witness_map_pre(self.a.data_ptr(), self.b.data_ptr(), ld, L, a_len=N + 2, b_len=N)   # ifft, ifft, coset_fft, coset_fft
        if self.scheme == "gsz":
            self._gsz_batch_mult()
        else:
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
        if self.base_split is None:
            ctx.msm_async(self.h_query, self.ab.data_ptr(), D, L, MONT, r["h"])
        else:
            self.h_q = self._sl(self.ab[:, :D - 1], D - 1)         # kept referenced until the MSM has read it
            ctx.msm_async(self.h_query, self.h_q.data_ptr(), self.h_q.shape[1], L, MONT, r["h"])
        if sync:
            ctx.sync()

    def _step_mac_from_sh(self, r, sync):
        """step() with the MSMs over the parties' sh lanes only (see mac_msm_from_sh); the witness map still runs every lane"""
        czk, ctx = self.czk, self.ctx
        D, N, L, ld = self.D, self.N, self.lanes, self.log_d
        M, ADD, MONT, P = czk.CZK_MEM_DEVICE, 0, czk.CZK_SCALAR_MONTGOMERY, self.lanes // 2
        ctx.msm_async(self.b_g2_query, self.asg_sh.data_ptr(), N + 1, P, MONT, r["b_g2"], stable=True)
        ctx.msm_async(self.l_query, self.wit_sh.data_ptr(), N, P, MONT, r["l"], stable=True)
        ctx.msm_async(self.a_query, self.asg_sh.data_ptr(), N + 1, P, MONT, r["a"], stable=True, same_scalars=True)
        ctx.msm_async(self.b_g1_query, self.asg_sh.data_ptr(), N + 1, P, MONT, r["b_g1"], stable=True, same_scalars=True)
        ctx.r1cs_matvec(self.mat_a, self.full.data_ptr(), lanes=L, out=self.a.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_b, self.full.data_ptr(), lanes=L, out=self.b.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.r1cs_matvec(self.mat_c, self.full.data_ptr(), lanes=L, out=self.c.data_ptr(), z_stride=N + 2, out_stride=D, mem=M)
        ctx.witness_map_pre(self.a.data_ptr(), self.b.data_ptr(), ld, L, a_len=N + 2, b_len=N)
        ctx.fr_vec_op(ADD, self.a.data_ptr(), self.tx.data_ptr(), out=self.a.data_ptr(), n=L * D, mem=M)
        ctx.fr_vec_op(ADD, self.b.data_ptr(), self.ty.data_ptr(), out=self.b.data_ptr(), n=L * D, mem=M)
        self._open(self.a, self.sx, self.chk[0])
        self._open(self.b, self.oy, self.chk[1])
        for ln in range(L):
            ctx.fr_beaver_combine(self.tx[ln].data_ptr(), self.ty[ln].data_ptr(), self.tz[ln].data_ptr(), self.sx.data_ptr(),
                                  self.oy.data_ptr(), ln in self.king_lanes, out=self.ab[ln].data_ptr(), n=D, mem=M)
        ctx.witness_map_post(self.ab.data_ptr(), self.c.data_ptr(), ld, L, c_len=N)
        self.ab_sh.copy_(self.ab[0::2])                      # the sh lanes of h, contiguous (torch's stream == the context's stream)
        ctx.msm_async(self.h_query, self.ab_sh.data_ptr(), D, P, MONT, r["h"])
        if sync:
            ctx.sync()

    def create_proof(self, res, r, s):
        """The rest of create_proof (mpc-snarks/src/groth/prover.rs:110-178) after the five MSMs `res` of step(): calculate_coeff (:216-232:
        `initial + query[0] + acc + vk_param`) for A, B in G1 and B in G2, and C = s A + r B1 - r s delta + l_acc + h_acc, for PUBLIC r, s
        (canonical (4,) uint64) -- the case that needs no group-level Beaver step: every operation is linear in the shares.  Public group
        elements meet a share through `shift` (MpcGroup Public + Shared: share/spdz.rs group shift, add.rs:141-146): the king adds them on
        its lanes (sh, and mac with the stand-in key 1), every other lane adds nothing.  Returns the lanes' shares of Proof{a, b, c} as
        Jacobian limbs {"a": (L, 18), "b": (L, 36), "c": (L, 18)}; summing the parties' sh lanes gives the proof itself.  Host-side O(1)
        group arithmetic through the C ABI (czk_jac_scalar_mul / czk_jac_add / czk_jac_add_mixed / czk_jac_neg)."""
        czk, ctx = self.czk, self.ctx
        G1, G2 = czk.CZK_G1, czk.CZK_G2
        res = self.expand_results(res)
        pk = self.pk
        r, s = np.ascontiguousarray(r, np.uint64).reshape(4), np.ascontiguousarray(s, np.uint64).reshape(4)

        def proj(group, aff):                                   # GroupAffine::into_projective
            one = to_mont_limbs_fq(1)
            return np.concatenate([aff, one if group == G1 else np.concatenate([one, np.zeros(6, np.uint64)])]).astype(np.uint64)
        delta_g1, delta_g2 = proj(G1, pk["delta_g1"]), proj(G2, pk["delta_g2"])
        r_s_delta_g1 = ctx.jac_scalar_mul(G1, ctx.jac_scalar_mul(G1, delta_g1, r), s)            # :113-117
        r_g1 = ctx.jac_scalar_mul(G1, delta_g1, r)                                               # :128
        s_g1 = ctx.jac_scalar_mul(G1, delta_g1, s)                                               # :144
        s_g2 = ctx.jac_scalar_mul(G2, delta_g2, s)                                               # :156

        def calculate_coeff(group, initial, el_aff, el_inf, acc, vk_param, king):               # :216-232, on one share lane
            if not king:                                        # Public + Shared = shift: only the king's lanes take the public addends
                return acc.copy()
            out = ctx.jac_add_mixed(group, initial, el_aff, el_inf)                              # res = initial; res.add_assign_mixed(&el)
            out = ctx.jac_add(group, out, acc)                                                   # res += &acc
            return ctx.jac_add_mixed(group, out, vk_param, False)                                # res.add_assign_mixed(&vk_param)
        L = res["h"].shape[0]
        out = {"a": np.zeros((L, 18), np.uint64), "b": np.zeros((L, 36), np.uint64), "c": np.zeros((L, 18), np.uint64)}
        inf1, inf2 = np.zeros(12, np.uint64), np.zeros(24, np.uint64)
        for ln in range(L):
            king = ln in self.king_lanes
            g_a = calculate_coeff(G1, r_g1, pk["a_query0"], False, res["a"][ln], pk["alpha_g1"], king)          # :135
            s_g_a = ctx.jac_scalar_mul(G1, g_a, s)                                                                # :138
            g1_b = calculate_coeff(G1, s_g1, inf1, True, res["b_g1"][ln], pk["beta_g1"], king)                   # :145
            g2_b = calculate_coeff(G2, s_g2, inf2, True, res["b_g2"][ln], pk["beta_g2"], king)                   # :157
            r_g1_b = ctx.jac_scalar_mul(G1, g1_b, r)                                                              # :158
            g_c = ctx.jac_add(G1, s_g_a, r_g1_b)                                                                  # :165-166
            if king:
                g_c = ctx.jac_add(G1, g_c, ctx.jac_neg(G1, r_s_delta_g1))                                         # g_c -= &r_s_delta_g1
            g_c = ctx.jac_add(G1, g_c, res["l"][ln])                                                              # :168
            g_c = ctx.jac_add(G1, g_c, res["h"][ln])                                                              # :169
            out["a"][ln], out["b"][ln], out["c"][ln] = g_a, g2_b, g_c
        return out

    def msm_scalars(self):
        """{query: device tensor of the Montgomery scalars its MSM consumed (lanes, n, 4)}; `h` is the LAST proof's."""
        return {"h": self.ab, "l": self.wit, "a": self.asg, "b_g1": self.asg, "b_g2": self.asg}

    def seam_calls_host_memory(self, reps: int = 1):
        """The NTT / MSM seam calls of one proof with every buffer in HOST memory (pageable, like a Rust Vec): 7
        czk_ntt_fr + 5 czk_msm per proof for all local lanes -- what an unmodified reference caller that binds only the
        two seams (INTEGRATION.md sections 2-3) pays, PCIe staging included.  Values are irrelevant to the timing."""
        czk, ctx = self.czk, self.ctx
        D, N, L, ld = self.D, self.N, self.lanes, self.log_d
        a = self.a0.cpu().numpy().view(np.uint64)
        wit = self.wit.cpu().numpy().view(np.uint64)
        asg = self.asg.cpu().numpy().view(np.uint64)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            for kind in (czk.CZK_IFFT, czk.CZK_IFFT, czk.CZK_COSET_FFT, czk.CZK_COSET_FFT, czk.CZK_IFFT, czk.CZK_COSET_FFT, czk.CZK_COSET_IFFT):
                ctx.ntt_fr(a, ld, kind, lanes=L)
            ctx.msm(self.h_query, a, n_scalars=D, lanes=L, scalar_form=czk.CZK_SCALAR_MONTGOMERY)
            ctx.msm(self.l_query, wit, n_scalars=N, lanes=L, scalar_form=czk.CZK_SCALAR_MONTGOMERY)
            for q in (self.a_query, self.b_g1_query, self.b_g2_query):
                ctx.msm(q, asg, n_scalars=N + 1, lanes=L, scalar_form=czk.CZK_SCALAR_MONTGOMERY)
        return (time.perf_counter() - t0) / reps

    def g1_accumulate_algorithmic_bytes(self):
        """an MSM of n points moves n*(96 B base) once + n*32 B of scalars per lane (n = this rank's base range)."""
        tot = 0
        for n in (self.D - 1, self.N, self.N + 1, self.N + 1):
            lo, hi = self.base_range(n)
            tot += (hi - lo) * 96 + self.lanes * (hi - lo) * 32
        return tot, 4   # bytes per step, launches per step

    def g1_mixed_additions_per_step(self, windows: int):
        return windows * self.lanes * sum(hi - lo for lo, hi in (self.base_range(n) for n in (self.D - 1, self.N, self.N + 1, self.N + 1)))

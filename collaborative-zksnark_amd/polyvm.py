"""Backend-neutral restatement of the LOCAL compute of the reference's polynomial-IOP provers (Plonk, Marlin) plus the GPU
backend that runs it through the C ABI.

The provers are written once against a small polynomial "machine" (`Backend`): arrays of Fr lanes with NTTs over radix-2 /
mixed-radix domains, element-wise arithmetic, multiplication by power tables, division by (X - z), running products, batch
inversion and KZG commitments (MSMs over a registered `powers_of_g`).  `GpuBackend` below is the product path; tests/ supplies a
second backend on top of the CPU checker and compares every commitment and evaluation the two produce.

What is restated: the call sequence and sizes of
    mpc-plonk/src/lib.rs:110-258 (prove_unit_product, prove_wiring), :259-340 (prove_public, prove_gates), :343-448 (eval, commit,
    prove);  marlin/src/ahp/prover.rs:300-704 (three AHP rounds), marlin/src/lib.rs:176-318 (commitments and openings)
What is NOT: protocol logic.  Fiat-Shamir challenges are fixed field elements (the hashing of commitments is host-side glue),
blinding factors are fixed, and a product of two shared vectors is the lane-wise product (for the reference's king_share
stand-in sharing every party holds the value itself, mpc-algebra/src/share/gsz20/mod.rs:190-213; the degree-reduction exchange
of `mult` is an open, covered by parallel.gsz_batch_open).
"""
from __future__ import annotations

import numpy as np

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
FFT, IFFT, COSET_FFT, COSET_IFFT = 0, 1, 2, 3


def mont(v: int) -> np.ndarray:
    """canonical integer -> (4,) uint64 Montgomery limbs"""
    x = v % R_MOD * ((1 << 256) % R_MOD) % R_MOD
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def unmont(limbs) -> int:
    x = sum(int(limbs[i]) << (64 * i) for i in range(4))
    return x * pow(1 << 256, -1, R_MOD) % R_MOD


def next_pow2(n: int) -> int:
    return 1 << max(0, (n - 1).bit_length())


def challenge(tag: str) -> int:
    """A fixed, documented stand-in for a Fiat-Shamir challenge: SHA-256(tag) mod r."""
    import hashlib
    return int.from_bytes(hashlib.sha256(tag.encode()).digest(), "little") % R_MOD


class Pending:
    """A commitment or an evaluation whose kernels are still in flight (GpuBackend): `.value` is filled by the next
    Backend.transcript_point()."""
    __slots__ = ("value", "_src")

    def __init__(self, src):
        self.value, self._src = None, src


class CommitmentSum:
    """`commitment.add_assign_mixed(&random_commitment)` (poly-commit/src/kzg10/mod.rs:188) / `w += ...` (:246-249): the O(1) group addition of
    two commitments (each possibly Pending), carried out on the host when the output is resolved."""
    __slots__ = ("B", "a", "b")

    def __init__(self, B, a, b):
        self.B, self.a, self.b = B, a, b


Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
_FQ_ONE = np.array([((1 << 384) % Q_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)   # Montgomery one of Fq


def resolved(x):
    """x with every Pending replaced by its value (dicts, lists and tuples are walked)."""
    if isinstance(x, CommitmentSum):
        return x.B.group_add(resolved(x.a), resolved(x.b))
    if isinstance(x, Pending):
        assert x.value is not None, "Pending read before a transcript_point()"
        return x.value
    if isinstance(x, dict):
        return {k: resolved(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(resolved(v) for v in x)
    return x


class Backend:
    """Arrays are (lanes, n, 4) Fr limbs (Montgomery).  Public data has lanes == 1.  Public * shared scales every lane; public
    +- shared is the reference's `shift`: the public operand is added on the lanes listed in `lift` only (GSZ: every lane, a
    share is an evaluation; SPDZ: the king's sh and mac lanes, share/spdz.rs:204-208 with mac_share() in {0, 1})."""
    lanes: int
    lift: tuple   # per lane: 1 = this lane takes public addends

    def zeros(self, lanes: int, n: int): raise NotImplementedError
    def lanes_of(self, a) -> int: raise NotImplementedError
    def lane_stack(self, parts): raise NotImplementedError           # list of (1, n, 4) arrays -> (len, n, 4)

    def lifted(self, a_public):
        """(lanes, n, 4): the public array on the lifting lanes, zero elsewhere."""
        n = self.length(a_public)
        z = self.zeros(1, n)
        return self.lane_stack([a_public if w else z for w in self.lift])

    def _addends(self, a, b):
        la, lb = self.lanes_of(a), self.lanes_of(b)
        if la == lb:
            return a, b
        return (self.lifted(a), b) if la == 1 else (a, self.lifted(b))

    # --- storage
    def upload(self, a: np.ndarray): raise NotImplementedError
    def download(self, a) -> np.ndarray: raise NotImplementedError
    def length(self, a) -> int: raise NotImplementedError
    def resized(self, a, n: int): raise NotImplementedError          # copy, zero-padded or truncated to n
    def drop_first(self, a, k: int): raise NotImplementedError       # coefficients k.. (copy)
    # --- transforms and arithmetic
    def ntt(self, a, size: int, kind: int): raise NotImplementedError   # new array of `size` elements
    def add(self, a, b): raise NotImplementedError                   # same lane count (use plus / minus for public operands)
    def sub(self, a, b): raise NotImplementedError
    def mul(self, a, b): raise NotImplementedError                   # broadcasts a public operand over the lanes
    def scale(self, a, k: int): raise NotImplementedError            # k: canonical integer
    def powers(self, g: int, n: int): raise NotImplementedError      # public (1, n, 4): g^i
    def div_linear(self, a, z: int): raise NotImplementedError       # (quotient array, remainder (lanes, 4) numpy)
    def prefix_product(self, a): raise NotImplementedError
    def inverse(self, a): raise NotImplementedError                  # element-wise, zeros stay zero
    def commit(self, a, key="g"): raise NotImplementedError          # -> ((lanes, 12) affine limbs, (lanes,) infinity flags) numpy; key "gamma": over powers_of_gamma_g
    def jac_add_mixed(self, a_jac, b_aff, b_inf): raise NotImplementedError   # host-side group step on (18,) / (12,) G1 limbs
    def jac_to_affine(self, jac): raise NotImplementedError          # (k, 18) -> ((k, 12), (k,))
    def random(self, seed: int, n: int): raise NotImplementedError   # public (1, n, 4): rand_fr_canonical(seed, n) in Montgomery form
    def root_of_unity(self, size: int) -> int: raise NotImplementedError   # get_root_of_unity(size), canonical integer
    def matrix(self, row_ptr, col, coeff, n_cols: int): raise NotImplementedError   # a public sparse matrix (CSR, Montgomery coefficients) for matvec
    def matvec(self, mat, v): raise NotImplementedError               # M v for a public (1, n_cols, 4) vector -> (1, rows, 4)

    # --- derived (shared by every backend) ----------------------------------------------------------------------
    def plus(self, a, b):
        return self.add(*self._addends(a, b))

    def minus(self, a, b):
        return self.sub(*self._addends(a, b))

    def const(self, k: int, n: int):
        """public array of n copies of k"""
        return self.upload(np.tile(mont(k), (n, 1)))

    def add_const(self, a, k: int):
        """k added to every element: for arrays of EVALUATIONS (`&evals + &F`, element-wise)"""
        return self.plus(a, self.const(k, self.length(a)))

    def poly_add_const(self, a, k: int):
        """`&DensePolynomial + &F` (algebra/poly/src/polynomial/univariate/dense.rs:301-315): a polynomial in COEFFICIENT form takes a constant on
        its coefficient 0 only."""
        return self.plus(a, self.resized(self.const(k, 1), self.length(a)))

    def poly_mul(self, a, b):
        """`&DensePolynomial * &DensePolynomial` (algebra/poly/src/polynomial/univariate/dense.rs): both operands are
        evaluated over GeneralEvaluationDomain::new(len a + len b - 1) -- always the radix-2 domain (domain/general.rs:168-181)
        -- multiplied point-wise and interpolated."""
        n = self.length(a) + self.length(b) - 1
        size = next_pow2(n)
        return self.resized(self.ntt(self.mul(self.ntt(a, size, FFT), self.ntt(b, size, FFT)), size, IFFT), n)

    def div_vanishing(self, a, n: int):
        """`divide_by_vanishing_poly` in coefficient form: (q, r) with a = q (X^n - 1) + r
        (algebra/poly/src/polynomial/univariate/dense.rs: q_i = sum_{j >= 1} a_{i + j n}, r_i = sum_{j >= 0} a_{i + j n})."""
        m = self.length(a)
        if m <= n:
            return self.resized(a, 0), a
        if n < (m + n - 1) // n:
            # few residues, many terms (Marlin divides by v_X with |X| = 2): the n residue classes a[c], a[c + n], ... are
            # independent polynomials in Y = X^n, and q's class c is the quotient of that polynomial by (Y - 1)
            L = (m + n - 1) // n
            classes = self.strided_split(self.resized(a, L * n), n)            # (lanes * n, L, 4)
            qc, rem = self.div_linear(classes, 1)
            lanes = self.lanes_of(a)
            q = self.strided_merge(self.resized(qc, L), n, lanes)               # back to (lanes, L * n, 4); the top n slots are zero
            r = self.upload(rem.reshape(lanes, n, 4))
            return self.resized(q, m - n), r
        chunks = [self.resized(self.drop_first(a, lo), n) for lo in range(0, m, n)]   # the last chunk is zero-padded
        suffix = chunks[-1]
        q_chunks = [None] * (len(chunks) - 1)
        for j in range(len(chunks) - 2, -1, -1):
            q_chunks[j] = suffix
            suffix = self.add(suffix, chunks[j])
        q = self.concat(q_chunks)
        return self.resized(q, m - n), suffix

    def concat(self, parts): raise NotImplementedError
    def strided_split(self, a, n: int): raise NotImplementedError    # (lanes, L n, 4) -> (lanes n, L, 4): class c of lane l at l n + c
    def strided_merge(self, a, n: int, lanes: int): raise NotImplementedError   # inverse of strided_split

    def shift(self, a, w: int):
        """mpc-plonk/src/util.rs:11-18: coefficient i times w^i."""
        return self.mul(a, self.powers(w, self.length(a)))

    def open_at(self, a, x: int, public=None):
        """`Prover::eval` (mpc-plonk/src/lib.rs:343-369) / KZG10::open (poly-commit/src/kzg10/mod.rs:225-265): the witness
        polynomial a / (X - x), its commitment, and the evaluation.  public: the polynomial is public data (its value needs no
        opening between the parties); None = infer from the lane count."""
        return self.open_finish(self.open_begin(a, x, public))

    def open_begin(self, a, x: int, public=None):
        """First half of open_at: the witness polynomial and the evaluation (KZG10::compute_witness_polynomial,
        poly-commit/src/kzg10/mod.rs:200-224); open_finish commits it.  Split because marlin_pc computes the witness of a
        degree-bounded polynomial before it opens the folded polynomial (marlin_pc/mod.rs:291-330)."""
        wit, value = self.div_linear(a, x)
        self.reveal(value)
        return {"value": value, "point": x, "_wit": wit}

    def open_finish(self, o):
        o["proof"] = self.commit(o.pop("_wit"))
        return o

    def quotient(self, a, x: int):
        """a / (X - x) without the remainder"""
        return self.div_linear(a, x)[0]

    def evaluate(self, a, x: int, public=None):
        """a(x) per lane, (lanes, 4) -- the remainder of the division by (X - x)"""
        return self.div_linear(a, x)[1]

    def group_add(self, ca, cb):
        """lane-wise sum of two commitments ((lanes, 12) affine limbs, (lanes,) infinity flags): GroupProjective::add_assign_mixed
        (short_weierstrass_jacobian.rs:570-638) on the host, then into_affine"""
        jacs = []
        for ln in range(ca[0].shape[0]):
            a_jac = np.concatenate([_FQ_ONE, _FQ_ONE, np.zeros(6, np.uint64)]) if ca[1][ln] else np.concatenate([ca[0][ln], _FQ_ONE])
            jacs.append(self.jac_add_mixed(a_jac, cb[0][ln], bool(cb[1][ln])))
        return self.jac_to_affine(np.stack(jacs))

    def transcript_point(self):
        """Called where the reference feeds commitments / evaluations to its Fiat-Shamir transcript before drawing the next
        challenge (mpc-plonk/src/lib.rs:110-113, 343-369; marlin/src/lib.rs:206-260): everything committed or evaluated so far
        must be final.  Backends that run asynchronously settle their pending results here."""
        return None

    def reveal(self, value):
        """`y.publicize()` of an evaluation (mpc-plonk/src/lib.rs:362-365; marlin/src/lib.rs:290): with one party per process the
        share of the value is opened over the network; with all lanes on one GPU there is nothing to exchange."""
        return None


def shared_stream_context(czk, device: int = 0):
    """A context whose kernels run on torch's current stream: GpuBackend mixes torch tensor operations (copies, concatenation,
    zero fills) with library calls, so both must be ordered on ONE stream (torch's default stream has handle 0, which the C ABI
    reads as "create a private stream": hence an explicit torch stream)."""
    import torch
    ts = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(ts)
    return czk.Context(device, ts.cuda_stream)


class GpuBackend(Backend):
    """The product path: device tensors, every operation one or a few C-ABI calls on the context's stream -- which must be
    torch's current stream (shared_stream_context)."""

    def __init__(self, czk, ctx, lanes: int, max_degree: int, base_seed: int = 0xBA5E5 + 77, lift=None, share_srs=None):
        import torch
        self.czk, self.ctx, self.lanes, self.torch = czk, ctx, lanes, torch
        self.lift = tuple([1] * lanes) if lift is None else tuple(lift)
        self.dev = torch.device("cuda")
        # powers_of_g = [tau^i] G for a FIXED, KNOWN tau (a real SRS hides tau; knowing it lets bench.py verify every opening on the
        # host without a pairing: C - [v] G == [tau - x] W).  k_i = tau^i as canonical scalars, computed on the device.
        n = max_degree + 1
        self.tau = challenge("kzg.tau.%x" % base_seed)
        if share_srs is not None:
            # a second prover on the same GPU (its own context and stream): registered bases are plain device data, shared
            assert share_srs.n_bases == n and share_srs.base_seed == base_seed
            self.bases, self.bases_host = share_srs.bases, share_srs.bases_host
        else:
            pw = torch.empty((n, 4), dtype=torch.int64, device=self.dev)
            ctx.fr_powers(mont(self.tau), n, out=pw.data_ptr(), mem=czk.CZK_MEM_DEVICE)
            k = torch.empty_like(pw)
            ctx.fr_into_repr(pw.data_ptr(), out=k.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
            pts = torch.empty((n, 12), dtype=torch.int64, device=self.dev)
            ctx.fixed_base_points(czk.CZK_G1, k.data_ptr(), out=pts.data_ptr(), n=n, mem=czk.CZK_MEM_DEVICE)
            self.bases = ctx.register_bases(czk.CZK_G1, pts.data_ptr(), None, n=n, mem=czk.CZK_MEM_DEVICE)
            ctx.sync()
            self.bases_host = (lambda: pts.cpu().numpy().view(np.uint64))      # for the checker-side backend of the tests
        # powers_of_gamma_g = [gamma tau^i] G (poly-commit/src/kzg10/mod.rs:92-101): the bases of the blinding polynomials' commitments; a hiding
        # bound of 1 needs three of them (data_structures.rs:464-481) -- eight are registered
        self.gamma = challenge("kzg.gamma.%x" % base_seed)
        if share_srs is not None:
            self.bases_gamma, self.bases_gamma_host = share_srs.bases_gamma, share_srs.bases_gamma_host
        else:
            kc = np.array([[(self.gamma * pow(self.tau, i, R_MOD) % R_MOD >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for i in range(8)], dtype=np.uint64)
            gpts = ctx.fixed_base_points(czk.CZK_G1, kc)                      # canonical scalars gamma tau^i
            self.bases_gamma = ctx.register_bases(czk.CZK_G1, gpts, None)
            self.bases_gamma_host = (lambda: gpts)
        self.base_seed, self.n_bases = base_seed, n
        self.msm_count = self.ntt_count = 0
        self._pending = []      # commitments / evaluations enqueued since the last transcript_point()
        self._vals, self._vals_at = None, 0      # evaluation buffer (_val_slot)
        self.opener = None      # party-per-rank layouts: callable(backend, (k, lanes, 4) device tensor) -> opened values, run over torch.distributed
        self.opened = []        # what the opener returned, in order
        self.msm_points = 0

    M = 1   # CZK_MEM_DEVICE

    def prepare(self, sizes):
        """czk_bases_prepare for the commitment lengths a prover will use: the narrower secondary table sets short polynomials run on are built at SRS
        load instead of inside the first proof's czk_msm_async (which would drain the MSM pipeline mid-proof to build them)."""
        for n in sorted(set(int(x) for x in sizes if 0 < int(x) <= self.n_bases)):
            self.bases.prepare(n)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        if a.ndim == 2:
            a = a[None]
        return self.torch.from_numpy(a.view(np.int64)).to(self.dev)

    def download(self, a):
        self.ctx.sync()
        return a.cpu().numpy().view(np.uint64)

    def length(self, a):
        return a.shape[1]

    def lanes_of(self, a):
        return a.shape[0]

    # ---- data movement: every re-layout between transforms is ONE czk_fr_copy_3d (or a copy-engine transfer when both sides are dense) on the
    # context's stream -- no tensor-library kernels in the provers' timelines (round 4: at::native fill / copy kernels were 45 % of the kernel
    # time inside Marlin's accumulate gaps).  torch only owns the allocations.  All arrays here are dense (lanes, n, 4) int64 tensors.
    def _new(self, lanes, n):
        return self.torch.empty((lanes, n, 4), dtype=self.torch.int64, device=self.dev)

    def _copy(self, dst, dst_off, dst_stride, src, src_off, src_stride, n3):
        """dst / src: tensors (src None: zero fill); offsets and strides in Fr elements"""
        if n3[0] * n3[1] * n3[2]:
            self.ctx.fr_copy_3d(dst.data_ptr() + 32 * dst_off, dst_stride, None if src is None else src.data_ptr() + 32 * src_off, src_stride, n3)

    def zeros(self, lanes, n):
        out = self._new(lanes, n)
        self._copy(out, 0, (0, n, 1), None, 0, None, (1, lanes, n))
        return out

    def lane_stack(self, parts):
        n = parts[0].shape[1]
        out = self._new(sum(p.shape[0] for p in parts), n)
        at = 0
        for p in parts:
            assert p.shape[1] == n
            self._copy(out, at * n, (0, n, 1), p, 0, (0, n, 1), (1, p.shape[0], n))
            at += p.shape[0]
        return out

    def resized(self, a, n):
        lanes, have = a.shape[0], a.shape[1]
        if n == have:
            return a                      # arrays are never written in place: the same array serves (`resize` to the current length is a no-op)
        out = self._new(lanes, n)
        m = min(n, have)
        self._copy(out, 0, (0, n, 1), a, 0, (0, have, 1), (1, lanes, m))
        self._copy(out, m, (0, n, 1), None, 0, None, (1, lanes, n - m))      # only the tail is cleared
        return out

    def lifted(self, a_public):
        """the public array on the lifting lanes, zero elsewhere: runs of equal lanes are one copy (source lane stride 0) or one fill each"""
        n = a_public.shape[1]
        out = self._new(self.lanes, n)
        at = 0
        while at < self.lanes:
            end = at
            while end < self.lanes and self.lift[end] == self.lift[at]:
                end += 1
            self._copy(out, at * n, (0, n, 1), a_public if self.lift[at] else None, 0, (0, 0, 1) if self.lift[at] else None, (1, end - at, n))
            at = end
        return out

    def drop_first(self, a, k):
        lanes, have = a.shape[0], a.shape[1]
        out = self._new(lanes, have - k)
        self._copy(out, 0, (0, have - k, 1), a, k, (0, have, 1), (1, lanes, have - k))
        return out

    def concat(self, parts):
        lanes, total = parts[0].shape[0], sum(p.shape[1] for p in parts)
        out = self._new(lanes, total)
        at = 0
        for p in parts:
            assert p.shape[0] == lanes
            self._copy(out, at, (0, total, 1), p, 0, (0, p.shape[1], 1), (1, lanes, p.shape[1]))
            at += p.shape[1]
        return out

    def strided_split(self, a, n):
        """out[l * n + j][k] = a[l][k * n + j]"""
        lanes, total = a.shape[0], a.shape[1]
        L = total // n
        out = self._new(lanes * n, L)
        self._copy(out, 0, (n * L, L, 1), a, 0, (total, 1, n), (lanes, n, L))
        return out

    def strided_merge(self, a, n, lanes):
        """out[l][k * n + j] = a[l * n + j][k]"""
        L = a.shape[1]
        out = self._new(lanes, L * n)
        self._copy(out, 0, (L * n, n, 1), a, 0, (n * L, 1, L), (lanes, L, n))
        return out

    def _bc(self, a, b):
        la, lb = a.shape[0], b.shape[0]
        if la != lb:
            one, many = (a, lb) if la == 1 else (b, la)
            n = one.shape[1]
            rep = self._new(many, n)
            self._copy(rep, 0, (0, n, 1), one, 0, (0, 0, 1), (1, many, n))      # lane stride 0: the single lane on every lane
            a, b = (rep, b) if la == 1 else (a, rep)
        return a, b

    def _vec(self, op, a, b):
        if op == 2:
            a, b = self._bc(a, b)
        assert a.shape == b.shape, (a.shape, b.shape)
        out = self._new(a.shape[0], a.shape[1])
        self.ctx.fr_vec_op(op, a.data_ptr(), b.data_ptr(), out=out.data_ptr(), n=a.shape[0] * a.shape[1], mem=self.M)
        return out

    def add(self, a, b):
        return self._vec(0, a, b)

    def sub(self, a, b):
        return self._vec(1, a, b)

    def mul(self, a, b):
        return self._vec(2, a, b)

    def scale(self, a, k):
        out = self._new(a.shape[0], a.shape[1])
        # the scalar travels with the launch (CZK_MEM_SCALAR_HOST): no staging slot, no 32-byte host-to-device copy per call
        self.ctx.fr_vec_scale(a.data_ptr(), mont(k), out=out.data_ptr(), n=a.shape[0] * a.shape[1], mem=self.M | self.czk.binding.CZK_MEM_SCALAR_HOST)
        return out

    def const(self, k, n):
        out = self._new(1, n)
        self.ctx.fr_powers(mont(1), n, c=mont(k), out=out.data_ptr(), mem=self.M)       # k * 1^i
        return out

    def powers(self, g, n):
        out = self.torch.empty((1, n, 4), dtype=self.torch.int64, device=self.dev)
        self.ctx.fr_powers(mont(g), n, out=out.data_ptr(), mem=self.M)
        return out

    def ntt(self, a, size, kind):
        m = min(a.shape[1], size)
        buf = self.torch.empty((a.shape[0], size, 4), dtype=self.torch.int64, device=self.dev)   # not zero-filled: the first pass zero-extends
        if size & (size - 1) == 0 and m > 0 and not self.ntt_copy_first:
            # radix-2 domain: the transform reads the source lanes itself (EvaluationDomain::fft(&coeffs) -> Vec: no copy of the operand)
            self.ctx.ntt_fr_to(a.data_ptr(), a.shape[1], buf.data_ptr(), size.bit_length() - 1, kind, lanes=buf.shape[0], in_len=m)
            self.ntt_count += buf.shape[0]
            return buf
        self._copy(buf, 0, (0, size, 1), a, 0, (0, a.shape[1], 1), (1, a.shape[0], m))          # beyond in_len itself (czk_ntt_fr_mixed)
        self.ctx.ntt_fr_mixed(buf.data_ptr(), size, kind, lanes=buf.shape[0], in_len=m, mem=self.M)
        self.ntt_count += buf.shape[0]
        return buf

    def _val_slot(self, rows):
        """`rows` Fr of the evaluation buffer: every value produced between two transcript points lands in ONE device array, so settling
        them is one device-to-host copy (no gather kernel)"""
        if self._vals is None:
            self._vals = self.torch.empty((4096, 4), dtype=self.torch.int64, device=self.dev)
        if self._vals_at + rows > self._vals.shape[0]:       # (a prover evaluates a few dozen values per round)
            # pending values are slices of the current buffer and transcript_point() reads them relative to ITS base: settle them before the buffer is
            # replaced (their values are final either way -- only the host copy happens earlier than the reference's transcript point)
            if self._pending:
                self.transcript_point()
            self.ctx.sync()
            self._vals = self.torch.empty((max(2 * self._vals.shape[0], rows), 4), dtype=self.torch.int64, device=self.dev)
            self._vals_at = 0
        v = self._vals[self._vals_at:self._vals_at + rows]
        self._vals_at += rows
        return v

    def _div_linear_dev(self, a, z):
        a = a.contiguous()
        lanes, n = a.shape[0], a.shape[1]
        q = self.torch.empty((lanes, max(n - 1, 0), 4), dtype=self.torch.int64, device=self.dev)
        rem = self._val_slot(lanes)
        self.ctx.poly_div_linear(a.data_ptr(), mont(z), lanes=lanes, n=n, quotient=q.data_ptr(), remainder=rem.data_ptr(), mem=self.M)
        return q, rem

    def div_linear(self, a, z):
        q, rem = self._div_linear_dev(a, z)
        self.ctx.sync()
        return q, rem.cpu().numpy().view(np.uint64)

    def quotient(self, a, x):
        return self._div_linear_dev(a, x)[0]

    def div_vanishing(self, a, n: int):
        """The suffix sums of a's n-coefficient chunks in ONE pass (czk_poly_div_vanishing) where the generic form makes a copy and an addition per chunk;
        few residues with long chains (Marlin's v_X, |X| = 2) keep the generic route through div_linear."""
        m = a.shape[1]
        if m <= n or n < (m + n - 1) // n:
            return super().div_vanishing(a, n)
        a = a.contiguous()
        lanes = a.shape[0]
        q = self.torch.empty((lanes, m - n, 4), dtype=self.torch.int64, device=self.dev)
        r = self.torch.empty((lanes, n, 4), dtype=self.torch.int64, device=self.dev)
        self.ctx.poly_div_vanishing(a.data_ptr(), n, lanes=lanes, m=m, quotient=q.data_ptr(), remainder=r.data_ptr(), mem=self.M)
        return q, r

    def _value_kind(self, a, public):
        if public is None:
            public = a.shape[0] == 1 and self.lanes > 1
        return "value" if public else "open_value"      # evaluations of share polynomials are publicized (mpc-plonk/src/lib.rs:362-365, marlin/src/lib.rs:283-292)

    def _evaluate_dev(self, a, x):
        a = a.contiguous()
        val = self._val_slot(a.shape[0])
        self.ctx.poly_evaluate(a.data_ptr(), mont(x), lanes=a.shape[0], n=a.shape[1], values=val.data_ptr(), mem=self.M)
        return val

    ntt_copy_first = False           # A/B switch (bench.py --ntt-copy-first): copy the operand, then transform in place, as rounds 2 - 3 did
    evaluate_by_division = False     # A/B switch (bench.py --eval-by-division): p(x) as the remainder of czk_poly_div_linear, as rounds 2 - 3 did

    def evaluate(self, a, x, public=None):
        p = Pending((self._value_kind(a, public), self._div_linear_dev(a, x)[1] if self.evaluate_by_division else self._evaluate_dev(a, x)))
        self._pending.append(p)
        return p

    def prefix_product(self, a):
        a = a.contiguous()
        out = self.torch.empty_like(a)
        for ln in range(a.shape[0]):
            self.ctx.fr_prefix_product(a[ln].data_ptr(), n=a.shape[1], out=out[ln].data_ptr(), mem=self.M)
        return out

    def inverse(self, a):
        a = a.contiguous()
        out = self.torch.empty_like(a)
        self.ctx.fr_batch_inverse(a.data_ptr(), n=a.shape[0] * a.shape[1], out=out.data_ptr(), mem=self.M)
        return out

    def random(self, seed, n):
        from .provers import rand_fr_canonical
        t = self.torch.from_numpy(rand_fr_canonical(seed, n).view(np.int64)).to(self.dev)
        out = self.torch.empty((1, n, 4), dtype=self.torch.int64, device=self.dev)
        self.ctx.fr_from_repr(t.data_ptr(), out=out.data_ptr(), n=n, mem=self.M)
        return out

    def root_of_unity(self, size):
        return unmont(self.ctx.mixed_domain_constants(size)["group_gen"])

    def jac_add_mixed(self, a_jac, b_aff, b_inf):
        return self.ctx.jac_add_mixed(self.czk.CZK_G1, a_jac, b_aff, b_inf)

    def jac_to_affine(self, jac):
        return self.ctx.jac_to_affine(self.czk.CZK_G1, jac)

    def matrix(self, row_ptr, col, coeff, n_cols):
        return self.ctx.r1cs_matrix_register(np.ascontiguousarray(row_ptr, dtype=np.uint64), np.ascontiguousarray(col, dtype=np.uint32),
                                             np.ascontiguousarray(coeff, dtype=np.uint64), n_cols), len(row_ptr) - 1, n_cols

    def matvec(self, mat, v):
        handle, rows, n_cols = mat
        v = v.contiguous()
        assert v.shape[0] == 1 and v.shape[1] == n_cols
        out = self._new(1, rows)
        self.ctx.r1cs_matvec(handle, v.data_ptr(), lanes=1, out=out.data_ptr(), z_stride=n_cols, out_stride=rows, mem=self.M)
        return out

    def commit(self, a, key="g"):
        """Enqueues the MSM (czk_msm_async: the sort / accumulate / reduce stages of consecutive commitments overlap on the
        library's streams, and with the NTTs enqueued after them) and returns a Pending; transcript_point() settles it."""
        a = a.contiguous()
        n = a.shape[1]
        bases = self.bases if key == "g" else self.bases_gamma
        assert n <= len(bases), "polynomial longer than the committer key"
        jac = np.zeros((a.shape[0], 18), dtype=np.uint64)
        self.ctx.msm_async(bases, a.data_ptr(), n_scalars=n, lanes=a.shape[0], scalar_form=self.czk.CZK_SCALAR_MONTGOMERY, out=jac)
        self.msm_count += a.shape[0]
        self.msm_points += a.shape[0] * n
        p = Pending(("commit", jac, a))       # `a` stays referenced until the MSM has read it
        self._pending.append(p)
        return p

    def open_begin(self, a, x, public=None):
        wit, rem = self._div_linear_dev(a, x)
        v = Pending((self._value_kind(a, public), rem))
        self._pending.append(v)
        return {"value": v, "point": x, "_wit": wit}

    def transcript_point(self):
        if not self._pending:
            if self._vals_at:
                self.ctx.sync()              # slots handed out for blocking reads (div_linear) may still be in flight
                self._vals_at = 0
            return
        self.ctx.sync()
        commits = [p for p in self._pending if p._src[0] == "commit"]
        if commits:                                                    # one conversion to affine for all of them
            aff, inf = self.ctx.jac_to_affine(self.czk.CZK_G1, np.concatenate([p._src[1] for p in commits]))
            at = 0
            for p in commits:
                k = p._src[1].shape[0]
                p.value = (aff[at:at + k].copy(), inf[at:at + k].copy())
                at += k
        values = [p for p in self._pending if p._src[0] != "commit"]
        if values:
            base = self._vals.data_ptr()
            host = self._vals[:self._vals_at].cpu().numpy().view(np.uint64)      # one copy: every slot handed out since the last settle
            for p in values:
                k, at = p._src[1].shape[0], (p._src[1].data_ptr() - base) // 32
                assert 0 <= at and at + k <= self._vals_at
                p.value = host[at:at + k].copy()
            opened = [p._src[1] for p in values if p._src[0] == "open_value"]
            if opened and self.opener is not None:
                # `y.publicize()` of every evaluation made since the last challenge, as ONE batch_open over the parties
                self.opened.append(self.opener(self, self.torch.stack(opened)))       # (k, lanes, 4)
        for p in self._pending:
            p._src = None
        self._pending = []
        self._vals_at = 0


# ---------------------------------------------------------------------------------------------------------------------
# Plonk (mpc-plonk/src/lib.rs)
# ---------------------------------------------------------------------------------------------------------------------
GENERATOR = 22   # Fr::multiplicative_generator() (fr.rs:69-74)


def vanishing(size: int, point: int) -> int:
    """evaluate_vanishing_polynomial: point^size - 1"""
    return (pow(point, size, R_MOD) - 1) % R_MOD


def shared_copy(B: Backend, a_public):
    """The same values on every lane: the reference's king_share stand-in hands every party the value itself
    (gsz20/mod.rs:190-213); for SPDZ benchmarks the lanes are filled the same way (values do not affect the work)."""
    return B.lane_stack([a_public] * B.lanes)


def plonk_inputs(B: Backend, n_gates: int, seed: int = 0x9107) -> dict:
    """A synthetic circuit layout of `n_gates` gates (relations/flat.rs:24-130): wire-value polynomial p (secret, 3 n_gates
    coefficients), selector s (public, n_gates), wiring permutation w (public, 3 n_gates), one public wire."""
    W = 3 * n_gates
    return {"n_gates": n_gates, "p": shared_copy(B, B.random(seed + 1, W)), "s": B.random(seed + 2, n_gates), "w": B.random(seed + 3, W)}


def plonk_prove(B: Backend, inp: dict) -> dict:
    """Local compute of `Prover::prove` (mpc-plonk/src/lib.rs:430-448) on plonk_inputs.  Returns every commitment and opening in
    the reference's order."""
    G = inp["n_gates"]
    W = 3 * G
    w = B.root_of_unity(W)                                                     # domains.wires.group_gen (mixed radix)
    zinv_w = pow(vanishing(W, GENERATOR), -1, R_MOD)                           # divide_by_vanishing_poly_on_coset (domain/mod.rs:184-191)
    out = {}
    p, s_pub, w_pub = inp["p"], inp["s"], inp["w"]

    def commit(label, a):
        out[label + "_cmt"] = B.commit(a)

    def open_(label, a, x, of=None):
        out[label] = B.open_at(a, x, public=of is None)
        out[label]["of"] = of          # label of the opened polynomial's commitment (None: an index polynomial, committed at setup)

    commit("p", p)                                                             # :434-441
    # prove_public (:259-292) with one public wire: v = p(x_pub) constant, z = X - x_pub, q = (p - v) / z
    q_pub = B.quotient(p, w)
    commit("pub_q", q_pub)
    B.transcript_point()
    x = challenge("plonk.public.x")
    open_("pub_q_open", q_pub, x, "pub_q")
    open_("pub_p_open", p, x, "p")
    # prove_gates (:295-340): d = s (p + pw) + (1 - s)(p pw) - pww, q = d / v_gates
    pw = B.shift(p, w)
    pww = B.shift(p, w * w % R_MOD)
    one_minus_s = B.poly_add_const(B.scale(s_pub, R_MOD - 1), 1)               # public: `&(&circ.s * &-F::one()) + &F::one()` (:307-308)
    d = B.sub(_padded_add(B, B.poly_mul(s_pub, B.add(p, pw)), B.poly_mul(one_minus_s, B.poly_mul(p, pw))), B.resized(pww, G + 2 * W - 2))
    q_gates, _r = B.div_vanishing(d, G)
    commit("gates_q", q_gates)
    B.transcript_point()
    x = challenge("plonk.gates.x")
    open_("gates_s_open", s_pub, x)
    open_("gates_p_open", p, x, "p")
    open_("gates_q_open", q_gates, x, "gates_q")
    open_("gates_p_w_open", p, w * x % R_MOD, "p")
    open_("gates_p_w2_open", p, w * w % R_MOD * x % R_MOD, "p")
    # prove_wiring (:201-257) over the wire domain
    B.transcript_point()
    y, z = challenge("plonk.wiring.y"), challenge("plonk.wiring.z")
    p_evals = B.ntt(p, W, FFT)
    w_evals = B.ntt(w_pub, W, FFT)
    yx_z = B.ntt(B.upload(np.stack([mont(z), mont(y)])), W, FFT)
    num_evals = B.add_const(B.plus(p_evals, B.scale(w_evals, y)), z)
    den_evals = B.plus(p_evals, yx_z)
    l1_evals = B.mul(num_evals, B.inverse(den_evals))
    l1 = B.ntt(l1_evals, W, IFFT)
    commit("l1", l1)
    # prove_unit_product(l1) (:115-198)
    t = B.ntt(B.prefix_product(B.ntt(l1, W, FFT)), W, IFFT)
    commit("t", t)
    f_c = B.ntt(B.shift(l1, w), W, COSET_FFT)
    t_c = B.ntt(t, W, COSET_FFT)
    tw_c = B.ntt(B.shift(t, w), W, COSET_FFT)
    q_up = B.ntt(B.scale(B.sub(tw_c, B.mul(f_c, t_c)), zinv_w), W, COSET_IFFT)
    commit("q", q_up)
    B.transcript_point()
    r = challenge("plonk.product.r")
    open_("t_wr_open", t, w * r % R_MOD, "t")
    open_("t_r_open", t, r, "t")
    open_("t_wk_open", t, pow(w, W - 1, R_MOD), "t")
    open_("f_wr_open", l1, w * r % R_MOD, "l1")
    open_("q_r_open", q_up, r, "q")
    # l2_q (:228-243)
    num_c, den_c = B.ntt(num_evals, W, IFFT), B.ntt(den_evals, W, IFFT)         # interpolate() of both before the coset transforms (:225-226)
    l1_v = B.ntt(l1, W, COSET_FFT)
    num_v = B.ntt(num_c, W, COSET_FFT)
    den_v = B.ntt(den_c, W, COSET_FFT)
    l2_q = B.ntt(B.scale(B.sub(B.mul(l1_v, den_v), num_v), zinv_w), W, COSET_IFFT)
    commit("l2_q", l2_q)
    B.transcript_point()
    x = challenge("plonk.wiring.x")
    open_("l2_q_x_open", l2_q, x, "l2_q")
    open_("w_x_open", w_pub, x)
    open_("l1_x_open", l1, x, "l1")
    open_("p_x_open", p, x, "p")
    B.transcript_point()
    return resolved(out)


def _padded_add(B, a, b):
    n = max(B.length(a), B.length(b))
    return B.plus(B.resized(a, n), B.resized(b, n))


def plonk_commit_sizes(n_gates: int):
    """lengths of the polynomials plonk_prove commits (for GpuBackend.prepare): p / l1 / t / q_up / l2_q (3 G), the public quotient and the opening
    witnesses (one shorter), gates_q (6 G - 2), the selector's witness (G - 1)"""
    G, W = n_gates, 3 * n_gates
    return [W, W - 1, 6 * G - 2, 6 * G - 3, G - 1]


def plonk_max_degree(n_gates: int) -> int:
    """Longest committed polynomial: gates_q has (G + 2 W - 2) - G = 6 G - 2 coefficients."""
    return 6 * n_gates


# ---------------------------------------------------------------------------------------------------------------------
# Marlin (marlin/src/ahp/prover.rs, marlin/src/lib.rs)
# ---------------------------------------------------------------------------------------------------------------------
def marlin_inputs(B: Backend, n_constraints: int, seed: int = 0x3A21) -> dict:
    """A synthetic index and assignment: |H| = next_pow2(n_constraints), |K| = next_pow2(non-zeros) with one non-zero per row
    and matrix (the reference's squaring circuit), two formatted inputs.  Witness-side vectors are share lanes; the arithmetised
    matrices (evaluations on K and on B, the index polynomials) are public index-time data (ahp/indexer.rs)."""
    H = K = next_pow2(n_constraints)
    X = 2
    b_size = next_pow2(3 * K - 3)
    inp = {"H": H, "K": K, "X": X, "b_size": b_size,
           "x": B.random(seed + 4, X),
           "w": shared_copy(B, B.concat([B.random(seed + 1, H - X), B.zeros(1, X)])),
           "z_a": shared_copy(B, B.random(seed + 2, H)), "z_b": shared_copy(B, B.random(seed + 3, H)),
           "mask_poly": shared_copy(B, B.random(seed + 5, 3 * H)),                # degree 3|H| + 2 zk - 3 with zk_bound = 1 (:376-380)
           "t_rows": B.random(seed + 6, H), "star": {}}
    for i, m in enumerate("abc"):
        inp["star"][m] = {"on_K": [B.random(seed + 10 * (i + 1) + j, K) for j in range(3)],              # row, col, val
                          "on_B": [B.random(seed + 10 * (i + 1) + 3 + j, b_size) for j in range(4)]}    # row, col, row_col, val
    inp["index_polys"] = [B.ntt(B.random(seed + 100 + j, K), K, IFFT) for j in range(12)]                 # row / col / val / row_col of A, B, C
    # ... committed at index time (marlin/src/lib.rs index(): the index_vk); the prover only re-uses the commitments
    cm = [B.commit(a) for a in inp["index_polys"]]
    B.transcript_point()
    inp["index_cmts"] = resolved(cm)
    return inp


def marlin_prove(B: Backend, inp: dict) -> dict:
    """Local compute of the three AHP prover rounds (marlin/src/ahp/prover.rs:300-704) and of Marlin::prove's commitments and
    openings (marlin/src/lib.rs:176-318) on marlin_inputs.  Witness-side polynomials are share lanes, the arithmetised matrices
    and everything in the third round are public, as in the reference."""
    H, K, X, b_size = inp["H"], inp["K"], inp["X"], inp["b_size"]
    out = {}
    blind = {}      # label -> blinding polynomial of a hiding commitment (share lanes, three coefficients)

    def commit(label, a, hiding=False):
        """marlin_pc::commit -> KZG10::commit (poly-commit/src/kzg10/mod.rs:141-192).  A hiding bound of Some(1) (prover.rs:386-388, :559) samples a
        blinding polynomial of degree hiding_bound + 1 (data_structures.rs:464-481), commits it over powers_of_gamma_g (:181-186) and adds the two
        commitments (:188).  (Without a hiding bound the reference still calls the second MSM with no coefficients: a no-op, not issued here.)  The
        blinding coefficients are fixed stand-ins, shared like every witness-side vector."""
        c = B.commit(a)
        if hiding:
            blind[label] = shared_copy(B, B.random(0xB11D + sum(map(ord, label)), 3))
            c = CommitmentSum(B, c, B.commit(blind[label], key="gamma"))
        out[label + "_cmt"] = c

    def mask(a, tag, n_dom):
        """a + rand * v_H: the zk blinding of the first-round polynomials (prover.rs:359-374); the blinding scalar is a fixed
        constant here.  rand * (X^n - 1) is added share-wise like any public polynomial."""
        rr = challenge("marlin.blind." + tag)
        bump = B.concat([B.const(R_MOD - rr, 1), B.zeros(1, n_dom - 1), B.const(rr, 1)])      # rr X^n - rr
        return B.plus(B.resized(a, n_dom + 1), bump)
    # ---- first round (prover.rs:300-398) -------------------------------------------------------------------
    x_poly = B.ntt(inp["x"], X, IFFT)                                            # public input polynomial (:324-330)
    x_evals = B.ntt(x_poly, H, FFT)
    w_evals = inp["w"]
    w_poly = B.ntt(B.minus(w_evals, x_evals), H, IFFT)                            # witness minus x on H, interpolated (:343-356)
    w_poly, _ = B.div_vanishing(mask(w_poly, "w", H), X)                          # / v_X (:357)
    z_a = mask(B.ntt(inp["z_a"], H, IFFT), "za", H)
    z_b = mask(B.ntt(inp["z_b"], H, IFFT), "zb", H)
    mask_poly = inp["mask_poly"]
    for label, a in (("w", w_poly), ("z_a", z_a), ("z_b", z_b), ("mask_poly", mask_poly)):
        commit(label, a, hiding=label != "mask_poly")                             # hiding bounds Some(1), Some(1), Some(1), None (:386-390)
    # ---- second round (:439-556) ----------------------------------------------------------------------------
    B.transcript_point()
    alpha, eta_a, eta_b, eta_c = (challenge("marlin." + t) for t in ("alpha", "eta_a", "eta_b", "eta_c"))
    z_c = B.poly_mul(z_a, z_b)                                                    # shared x shared (:466)
    summed = _padded_add(B, B.scale(z_c, eta_c), B.add(B.scale(z_a, eta_a), B.scale(z_b, eta_b)))   # (:468-476)
    # r(alpha, X) on H, unnormalised bivariate Lagrange: (alpha^|H| - 1) / (alpha - h^i)  (:480-482), public
    hpow = B.powers(B.root_of_unity(H), H)
    r_alpha_evals = B.scale(B.inverse(B.add_const(B.scale(hpow, R_MOD - 1), alpha)), vanishing(H, alpha))
    r_alpha_poly = B.ntt(r_alpha_evals, H, IFFT)
    if inp.get("matrices_T") is not None:
        # calculate_t (:400-416) on a real index: t_evals[reindex(c)] += eta_M M[r][c] r_alpha[r] -- (sum_M eta_M M~)^T r_alpha with the column positions
        # re-indexed by the input sub-domain; `matrices_T` holds the transposed, re-indexed matrices
        t_evals = None
        for m, eta in (("a", eta_a), ("b", eta_b), ("c", eta_c)):
            term = B.scale(B.matvec(inp["matrices_T"][m], r_alpha_evals), eta)
            t_evals = term if t_evals is None else B.add(t_evals, term)
        t_poly = B.ntt(t_evals, H, IFFT)
    else:
        t_poly = B.ntt(B.mul(inp["t_rows"], r_alpha_evals), H, IFFT)      # the benchmark's index: one weight per row (same work: one pass over |H| elements and a transform)
    x_poly = B.ntt(inp["x"], X, IFFT)                                              # interpolated again in the second round (:503-507)
    z_poly = _padded_add(B, _mul_by_vanishing(B, w_poly, X), x_poly)              # w v_X + x (:512-517)
    n_rhs = max(B.length(r_alpha_poly) + B.length(summed), B.length(t_poly) + B.length(z_poly)) - 1
    mul_size = next_pow2(max(B.length(mask_poly), n_rhs + 1))                     # GeneralEvaluationDomain::new(max(..)) (:522-531)
    ev = lambda a: B.ntt(a, mul_size, FFT)
    rhs = B.resized(B.ntt(B.sub(B.mul(ev(r_alpha_poly), ev(summed)), B.mul(ev(z_poly), ev(t_poly))), mul_size, IFFT), n_rhs)
    q_1 = _padded_add(B, mask_poly, rhs)
    h_1, x_g_1 = B.div_vanishing(q_1, H)
    g_1 = B.drop_first(x_g_1, 1)
    for label, a in (("t", t_poly), ("g_1", g_1), ("h_1", h_1)):
        commit(label, a, hiding=label == "g_1")                                    # hiding bounds None, Some(1), None (:558-560)
        if label == "g_1":
            # a degree-bounded oracle (g_1: |H| - 2) carries a second commitment over the shifted powers (marlin_pc/mod.rs commit: `shifted_comm`):
            # the same scalars, the same length -- the stand-in key commits over the same prefix of powers -- with its own blinding polynomial
            # (`shifted_rand`, marlin_pc/mod.rs:218-232)
            commit("g_1_shifted", a, hiding=True)
    # ---- third round (:585-704): everything public ---------------------------------------------------------------
    B.transcript_point()
    beta = challenge("marlin.beta")
    vh = vanishing(H, alpha) * vanishing(H, beta) % R_MOD
    etas = {"a": eta_a, "b": eta_b, "c": eta_c}
    f_evals, den_b, val_b = None, {}, {}
    for i, m in enumerate("abc"):
        row, col, val = inp["star"][m]["on_K"]                                     # a_star.evals_on_K.{row, col, val}
        inv = B.inverse(B.mul(B.add_const(B.scale(row, R_MOD - 1), beta), B.add_const(B.scale(col, R_MOD - 1), alpha)))   # (:612-620)
        term = B.scale(B.mul(val, inv), etas[m])
        f_evals = term if f_evals is None else B.add(f_evals, term)
        rb, cb, rcb, vb = inp["star"][m]["on_B"]                                    # a_star.evals_on_B.*, row_col_evals_on_B
        # beta alpha - r alpha - beta c + r_c  (:641-658)
        den_b[m] = B.add_const(B.add(B.sub(rcb, B.scale(rb, alpha)), B.scale(cb, R_MOD - beta)), beta * alpha % R_MOD)
        val_b[m] = vb
    f = B.ntt(B.scale(f_evals, vh), K, IFFT)
    g_2 = B.drop_first(f, 1)
    a_on_b = None
    for m, o1, o2 in (("a", "b", "c"), ("b", "a", "c"), ("c", "a", "b")):
        term = B.scale(B.mul(val_b[m], B.mul(den_b[o1], den_b[o2])), etas[m])      # (:664-673)
        a_on_b = term if a_on_b is None else B.add(a_on_b, term)
    # in a real index row, col, val and row_col have degree |K| - 1, so a and b have degree 3 |K| - 3: 3 |K| - 2 coefficients (domain_b is asked for 3 |K| - 3
    # points, :636, and holds them because it rounds up to a power of two); keep that many of the stand-in data's
    a_poly = B.resized(B.ntt(B.scale(a_on_b, vh), b_size, IFFT), 3 * K - 2)
    b_poly = B.resized(B.ntt(B.mul(den_b["a"], B.mul(den_b["b"], den_b["c"])), b_size, IFFT), 3 * K - 2)
    bf = B.poly_mul(b_poly, f)
    h_2, _ = B.div_vanishing(B.sub(B.resized(a_poly, B.length(bf)), bf), K)        # (a - b f) / v_K (:693-696)
    for label, a in (("g_2", g_2), ("h_2", h_2)):
        commit(label, a)
        if label == "g_2":
            commit("g_2_shifted", a)                                                # degree bound |K| - 2
    # ---- evaluations and openings (marlin/src/lib.rs:262-318) ---------------------------------------------------------------
    B.transcript_point()
    gamma = challenge("marlin.gamma")
    idx = inp["index_polys"]                                                       # row, col, val, row_col of A, B, C
    row, col, val, row_col = ({m: idx[4 * i + j] for i, m in enumerate("abc")} for j in range(4))
    polys = {"w": w_poly, "z_a": z_a, "z_b": z_b, "mask_poly": mask_poly, "t": t_poly, "g_1": g_1, "h_1": h_1, "g_2": g_2, "h_2": h_2}
    for i, m in enumerate("abc"):
        polys.update({m + "_row": row[m], m + "_col": col[m], m + "_val": val[m], m + "_row_col": row_col[m]})
        for j, part in enumerate(("row", "col", "val", "row_col")):
            out[m + "_" + part + "_cmt"] = inp["index_cmts"][4 * i + j]            # index-time commitments (not prover work)
    # the linear combinations of AHPForR1CS::construct_linear_combinations (marlin/src/ahp/mod.rs:115-260); their coefficients are
    # products of challenges and evaluations in the reference -- fixed stand-ins here (values do not change the work); constant
    # (LCTerm::One) terms do not enter the opened polynomial (poly-commit/src/marlin/mod.rs:256)
    lcs = {"z_b": [(1, "z_b")], "g_1": [(1, "g_1")], "t": [(1, "t")], "g_2": [(1, "g_2")],
           "outer_sumcheck": [(1, "mask_poly"), (challenge("marlin.lc.z_a"), "z_a"), (challenge("marlin.lc.w"), "w"), (challenge("marlin.lc.h_1"), "h_1")],
           "inner_sumcheck": [(challenge("marlin.lc." + m + "_val"), m + "_val") for m in "abc"] + [(challenge("marlin.lc.h_2"), "h_2")]}
    for m in "abc":
        lcs[m + "_denom"] = [(R_MOD - alpha, m + "_row"), (R_MOD - beta, m + "_col"), (1, m + "_row_col")]
    point = {"beta": beta, "gamma": gamma}
    query = {"beta": ["g_1", "outer_sumcheck", "t", "z_b"], "gamma": ["a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"]}   # verifier_query_set (ahp/verifier.rs:143-146, 207-211), labels in BTreeSet order

    def lc_eval(label, tag):
        """EvaluationsProvider::get_lc_eval for the prover's polynomials (ahp/mod.rs:288-312): every polynomial of the combination
        is evaluated at the point (Polynomial::evaluate), the sum is publicized"""
        for _, name in lcs[label]:
            out.setdefault("evals_" + tag, []).append(B.evaluate(polys[name], point[tag]))
    # construct_linear_combinations evaluates what the coefficients of the two sumcheck combinations need (:155-157, :228-231) ...
    for label, tag in (("z_b", "beta"), ("t", "beta"), ("g_1", "beta"), ("a_denom", "gamma"), ("b_denom", "gamma"), ("c_denom", "gamma"), ("g_2", "gamma")):
        lc_eval(label, tag)
    # ... and Marlin::prove evaluates every queried combination (lib.rs:283-292; the query set iterates in label order)
    for label in sorted(lcs):
        lc_eval(label, "beta" if label in query["beta"] else "gamma")
    B.transcript_point()                                                           # fs_rng.absorb(&evaluations) (:299)
    if inp.get("real_lcs"):
        # The coefficients AHPForR1CS::construct_linear_combinations (ahp/mod.rs:115-260) really uses: products of challenges and of evaluations the prover has
        # just publicized (:155-157, :228-231) -- the first entries of the two evaluation lists, settled by the transcript point above like everything else, so
        # the real path has the benchmark's synchronisation points and no others.  Lane 0 (every lane is the plain prover when the caller lifts public data onto
        # all of them).  The LCTerm::One constants go to out["lc_consts"]: they do not enter the opened polynomials, the verifier moves them to the evaluation side.
        val = lambda pend: unmont(resolved(pend)[0])            # noqa: E731
        inv = lambda v: pow(v % R_MOD, -1, R_MOD)                # noqa: E731
        eb, eg = out["evals_beta"], out["evals_gamma"]
        z_b_beta, t_beta, g_1_beta = val(eb[0]), val(eb[1]), val(eb[2])
        den = {}
        for i, m in enumerate("abc"):                           # a_denom = beta alpha - alpha row - beta col + row_col at gamma (:196-226)
            r_g, c_g, rc_g = (val(eg[3 * i + j]) for j in range(3))
            den[m] = (beta * alpha - alpha * r_g - beta * c_g + rc_g) % R_MOD
        g_2_gamma = val(eg[9])
        wx, x_beta, acc = B.root_of_unity(X), 0, 1             # x(beta) = sum_j L_j(beta) x_j over the input domain (:158-163)
        for xj in inp["x_ints"]:
            x_beta = (x_beta + vanishing(X, beta) * acc % R_MOD * inv(X * (beta - acc)) % R_MOD * xj) % R_MOD
            acc = acc * wx % R_MOD
        r_ab = (vanishing(H, alpha) - vanishing(H, beta)) * inv(alpha - beta) % R_MOD          # eval_unnormalized_bivariate_lagrange_poly
        vh_a, vh_b, vx_b = vanishing(H, alpha), vanishing(H, beta), vanishing(X, beta)
        lcs["outer_sumcheck"] = [(1, "mask_poly"), (r_ab * (eta_a + eta_c * z_b_beta) % R_MOD, "z_a"), (-t_beta * vx_b % R_MOD, "w"), (-vh_b % R_MOD, "h_1")]
        consts = {"outer_sumcheck": (r_ab * eta_b % R_MOD * z_b_beta - t_beta * x_beta - beta * g_1_beta) % R_MOD}
        lcs["inner_sumcheck"] = [(eta_a * den["b"] % R_MOD * den["c"] % R_MOD * vh_a % R_MOD * vh_b % R_MOD, "a_val"),
                                 (eta_b * den["a"] % R_MOD * den["c"] % R_MOD * vh_a % R_MOD * vh_b % R_MOD, "b_val"),
                                 (eta_c * den["b"] % R_MOD * den["a"] % R_MOD * vh_a % R_MOD * vh_b % R_MOD, "c_val"),
                                 (-vanishing(K, gamma) % R_MOD, "h_2")]
        consts["inner_sumcheck"] = -(den["a"] * den["b"] % R_MOD * den["c"] % R_MOD) * (gamma * g_2_gamma + t_beta * inv(K)) % R_MOD
        for m in "abc":
            consts[m + "_denom"] = beta * alpha % R_MOD
        out["lc_consts"] = consts
        out["lcs"] = {k: list(v) for k, v in lcs.items()}
    ch = challenge("marlin.opening_challenge")
    # PC::open_combinations (poly-commit/src/marlin/mod.rs:213-300): one polynomial per combination ...
    lc_poly = {}
    for label, terms in lcs.items():
        acc = None
        for coef, name in terms:
            term = polys[name] if coef == 1 else B.scale(polys[name], coef)
            acc = term if acc is None else _padded_add(B, acc, term)
        lc_poly[label] = acc
    # ... then batch_open (poly-commit/src/lib.rs:597-640): per query point the queried polynomials are folded with powers of the opening
    # challenge and opened once (marlin_pc/mod.rs:259-316); a degree-bounded polynomial (g_1: |H| - 2, g_2: |K| - 2) takes two
    # challenges and also opens its own witness polynomial over the shifted powers (:291-310, :318-330)
    for tag in ("beta", "gamma"):
        folded, c, terms = None, 1, []
        shifted = None
        for label in query[tag]:
            a = lc_poly[label]
            term = a if c == 1 else B.scale(a, c)
            folded = term if folded is None else _padded_add(B, folded, term)
            terms += [(c * coef % R_MOD, name) for coef, name in lcs[label]]
            c = c * ch % R_MOD
            if label in ("g_1", "g_2"):
                shifted = label
                c = c * ch % R_MOD
        # the shifted-witness opening: (g - g(point)) / (X - point), computed inside the folding loop (:294-299), committed over the
        # shifted powers after the folded polynomial has been opened (:318-330) -- the stand-in key commits over the same prefix of
        # powers (same scalars, same length: same work)
        # Inside the folding loop the reference computes, for the degree-bounded polynomial, its witness AND the witness of its shifted randomness
        # (compute_witness_polynomial(polynomial, point, shifted_rand), marlin_pc/mod.rs:294-299 -> kzg10/mod.rs:200-224) ...
        sh = B.open_begin(polys[shifted], point[tag], public=B.lanes_of(polys[shifted]) == 1 and B.lanes > 1)
        sh_rand = B.open_begin(blind[shifted + "_shifted"], point[tag]) if shifted + "_shifted" in blind else None
        # ... then opens the folded polynomial with the folded randomness (`r += (challenge_j, &rand.rand)`, :288; KZG10::open :313): witness and
        # random witness r / (X - point) (kzg10 :211-218), MSM of the witness, blinding evaluation r(point), MSM of the random witness over
        # powers_of_gamma_g added to w (:238-249).  Only beta's query set holds hiding polynomials (w, z_a, z_b, g_1).
        r_fold = None
        for coef, name in terms:
            if name in blind:
                term = blind[name] if coef == 1 else B.scale(blind[name], coef)
                r_fold = term if r_fold is None else B.add(r_fold, term)
        o = B.open_begin(folded, point[tag])
        rw = B.open_begin(r_fold, point[tag]) if r_fold is not None else None
        o = B.open_finish(o)
        o["terms"] = terms                                                          # the opened polynomial as sum coef * committed polynomial
        if rw is not None:
            o["random_v"] = rw["value"]
            o["proof"] = CommitmentSum(B, o["proof"], B.commit(rw.pop("_wit"), key="gamma"))
        out["open_" + tag] = o
        # ... and the shifted witness over the shifted powers, with the shifted randomness' witness (open_with_witness_polynomial, :318-330)
        so = B.open_finish(sh)
        so["of"] = shifted + "_shifted" if sh_rand is not None else shifted
        if sh_rand is not None:
            so["random_v"] = sh_rand["value"]
            so["proof"] = CommitmentSum(B, so["proof"], B.commit(sh_rand.pop("_wit"), key="gamma"))
        out["open_" + tag + "_shifted"] = so
    B.transcript_point()
    return resolved(out)


def _mul_by_vanishing(B, a, n):
    """a * (X^n - 1) (dense.rs mul_by_vanishing_poly): a shifted up by n, minus a"""
    m = B.length(a)
    return B.sub(B.concat([B.zeros(B.lanes_of(a), n), a]), B.resized(a, m + n))


def marlin_commit_sizes(n_constraints: int):
    """lengths of the polynomials marlin_prove commits (for GpuBackend.prepare)"""
    H = K = next_pow2(n_constraints)
    return [H - 1, H + 1, 3 * H, H, H - 1, 2 * H, K - 1, 3 * K - 3, 3 * H - 1, H - 2, 3 * K - 4, K - 2, K]


def marlin_max_degree(n_constraints: int) -> int:
    """Longest committed polynomial: mask_poly (3 |H| coefficients) or h_2 = (a - b f) / v_K (3 |K| - 3)."""
    return 3 * next_pow2(n_constraints) + 8

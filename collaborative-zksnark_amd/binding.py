"""ctypes binding of libczk_hip.so (include/czk.h).  Host arrays are numpy uint64; device arrays are raw
pointers (ints) -- e.g. `tensor.data_ptr()` of a torch int64/uint64 CUDA tensor."""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("CZK_LIB_PATH") or os.path.join(_HERE, "libczk_hip.so")   # CZK_LIB_PATH: A/B builds of the library (tools/)
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "czk.h")

CZK_FFT, CZK_IFFT, CZK_COSET_FFT, CZK_COSET_IFFT = 0, 1, 2, 3
CZK_MEM_HOST, CZK_MEM_DEVICE = 0, 1
CZK_MEM_NO_TABLES = 32   # czk_bases_register: OR-ed with the above, see include/czk.h
CZK_MEM_ANY_POINTS = 64  # czk_bases_register: bases need not lie in the prime-order subgroup (keeps the XYZZ kernels for G1)
CZK_MEM_SCALAR_HOST = 256    # czk_fr_vec_scale: device vectors, host scalar
CZK_MEM_SAME_SCALARS = 512   # czk_msm_async: the scalars of the previous czk_msm_async call (its digit sort may be reused)
CZK_MEM_CHECK_SUBGROUP = 128  # czk_bases_register: verify [r] P == infinity; a failing base keeps the handle on the XYZZ kernels
CZK_SCALAR_CANONICAL, CZK_SCALAR_MONTGOMERY = 0, 1
CZK_G1, CZK_G2 = 1, 2
CZK_OP_ADD, CZK_OP_SUB, CZK_OP_MUL = 0, 1, 2
CZK_NET_RCCL, CZK_NET_SHM, CZK_NET_IPC = 1, 2, 3
CZK_OPEN_COMMIT = 1
CZK_ERR_NET, CZK_ERR_CHECK = 5, 6
_STATUS = {1: "CZK_ERR_SIZE", 2: "CZK_ERR_HIP", 3: "CZK_ERR_ARG", 4: "CZK_ERR_NOMEM", 5: "CZK_ERR_NET", 6: "CZK_ERR_CHECK"}


class CzkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_STATUS.get(code, code)}: {msg}")
        self.code = code


def lib_path() -> str:
    return _LIB


_LIB_LAB = os.path.join(_HERE, "libczk_hip_lab.so")   # the -DCZK_LAB build: product + the rejected variants kept for A/B runs (build.py)
_lib = None
_lab = None


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback for the product path)")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes cannot share a process, so let
    # torch's copy load first (same SONAME -> the library binds to it).  torch is plumbing here: device
    # memory, streams, torch.distributed.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see csrc/core.hip: must be set before HIP initialises
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.czk_last_error.restype = C.c_char_p
    L.czk_version.restype = C.c_char_p
    L.czk_bases_len.restype = C.c_size_t
    L.czk_lanes_count.restype = C.c_size_t
    L.czk_lanes_len.restype = C.c_size_t
    L.czk_lanes_data.restype = C.c_void_p
    L.czk_ctx_stream.restype = C.c_void_p
    L.czk_net_last_error.restype = C.c_char_p
    L.czk_net_destroy.restype = None
    L.czk_net_stats_reset.restype = None
    L.czk_sha256.restype = None
    return L


def lab_lib():
    """The lab build (libczk_hip_lab.so): same ABI, plus the rejected kernel variants and their options.  Tests and tools only."""
    global _lab
    if _lab is None:
        _lab = _load(_LIB_LAB)
    return _lab


def lib():
    """Loads libczk_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(_LIB)
    return _lib


def header_symbols() -> list[str]:
    """Every function include/czk.h declares."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(czk_[a-z0-9_]+)\s*\(", txt)))


def exported_symbols() -> list[str]:
    l = lib()
    return [s for s in header_symbols() if hasattr(l, s)]


def _ptr(x):
    """numpy array -> void*, int -> void*, None -> NULL"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(x))


class Context:
    """czk_ctx: one GPU + one HIP stream = one MPC party.

    `stream` is a hipStream_t handle (e.g. `torch.cuda.Stream().cuda_stream`).  None / 0 makes the context create its
    own non-blocking stream -- note torch's DEFAULT stream has handle 0, so to share a stream with torch ops use an
    explicit `torch.cuda.Stream()` (bench.py does)."""

    def __init__(self, device: int = 0, stream: int | None = None, lab: bool = False, options: dict | None = None):
        """lab=True binds this context to libczk_hip_lab.so (tests / A/B tools: the rejected kernel variants and their options);
        `options`: {name: value} passed to czk_ctx_set_option before any work."""
        self._L = lab_lib() if lab else lib()
        self._h = C.c_void_p(0)
        rc = self._L.czk_ctx_create(C.byref(self._h), C.c_int(device), C.c_void_p(stream or 0))
        if rc:
            raise CzkError(rc, "czk_ctx_create failed (no visible GPU?)")
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def stream_handle(self) -> int:
        """hipStream_t of this context as an integer (czk_ctx_stream): torch.cuda.ExternalStream(handle) orders torch work against it"""
        return int(self._L.czk_ctx_stream(self._h) or 0)

    def reserve(self, ntt_log_d: int = 0, ntt_lanes: int = 0, bases=None, n_scalars: int = 0, msm_lanes: int = 0):
        """czk_ctx_reserve: build NTT tables / size MSM workspaces now instead of inside the first proof"""
        self._ck(self._L.czk_ctx_reserve(self._h, C.c_uint(ntt_log_d), C.c_size_t(ntt_lanes), bases._h if bases is not None else None, C.c_size_t(n_scalars),
                                         C.c_size_t(msm_lanes)))

    def set_option(self, name: str, value: int):
        self._ck(self._L.czk_ctx_set_option(self._h, name.encode(), C.c_long(int(value))))

    def close(self):
        if self._h:
            self._L.czk_ctx_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise CzkError(rc, (self._L.czk_last_error(self._h) or b"").decode())

    def sync(self):
        self._ck(self._L.czk_ctx_sync(self._h))

    def mark(self) -> int:
        """czk_ctx_mark: names the work enqueued on the context so far (kernels, msm_async calls, deferred downloads)"""
        m = C.c_uint64(0)
        self._ck(self._L.czk_ctx_mark(self._h, C.byref(m)))
        return m.value

    def wait_mark(self, mark: int):
        """czk_ctx_wait_mark: blocks until the work before `mark` is done and delivers its host results; later calls keep running"""
        self._ck(self._L.czk_ctx_wait_mark(self._h, C.c_uint64(mark)))

    # ---- device-resident share lanes ---------------------------------------------------------------
    def lanes_alloc(self, lanes: int, length: int) -> "Lanes":
        """czk_lanes_alloc: `lanes` x `length` Fr in HBM, zero-filled; the handle a caller without a HIP allocator uses."""
        h = C.c_void_p(0)
        self._ck(self._L.czk_lanes_alloc(self._h, C.c_size_t(lanes), C.c_size_t(length), C.byref(h)))
        return Lanes(self, h)

    # ---- NTT ------------------------------------------------------------------------------------
    def ntt_fr_to(self, src, src_stride: int, dst, log_d: int, kind: int, lanes: int = 1, in_len: int | None = None):
        """EvaluationDomain::{fft,ifft,coset_fft,coset_ifft} (not in place): device pointers; reads in_len elements per source lane (stride
        src_stride elements), writes 2^log_d per lane to dst."""
        if in_len is None:
            in_len = min(src_stride, 1 << log_d)
        self._ck(self._L.czk_ntt_fr_to(self._h, _ptr(src), C.c_size_t(src_stride), _ptr(dst), C.c_uint(log_d), C.c_size_t(lanes), C.c_int(kind),
                                     C.c_size_t(in_len), C.c_int(CZK_MEM_DEVICE)))
        return dst

    def ntt_fr(self, data, log_d: int, kind: int, lanes: int = 1, in_len: int | None = None, mem: int = CZK_MEM_HOST):
        """EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place on `lanes` Fr lanes (in place)."""
        d = 1 << log_d
        if in_len is None:
            in_len = d
        if isinstance(data, np.ndarray):
            assert data.dtype == np.uint64 and (log_d > 40 or data.size == lanes * d * 4), "buffer must hold lanes x D x 4 u64"
        self._ck(self._L.czk_ntt_fr(self._h, _ptr(data), C.c_uint(log_d), C.c_size_t(lanes), C.c_int(kind), C.c_size_t(in_len),
                                  C.c_int(mem)))
        return data

    def ntt_fr_mixed(self, data, size: int, kind: int, lanes: int = 1, in_len: int | None = None, mem: int = CZK_MEM_HOST):
        """MixedRadixEvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place over a domain of `size` = 2^a or 3 * 2^a."""
        if in_len is None:
            in_len = size
        if isinstance(data, np.ndarray):
            assert data.dtype == np.uint64 and data.size == lanes * size * 4, "buffer must hold lanes x size x 4 u64"
        self._ck(self._L.czk_ntt_fr_mixed(self._h, _ptr(data), C.c_size_t(size), C.c_size_t(lanes), C.c_int(kind), C.c_size_t(in_len), C.c_int(mem)))
        return data

    def mixed_domain_constants(self, size: int):
        out = np.zeros((6, 4), dtype=np.uint64)
        self._ck(self._L.czk_mixed_domain_constants(self._h, C.c_size_t(size), _ptr(out)))
        names = ["size_inv", "group_gen", "group_gen_inv", "generator", "generator_inv", "vanishing_inv"]
        return dict(zip(names, out))

    def domain_constants(self, log_d: int):
        out = np.zeros((6, 4), dtype=np.uint64)
        self._ck(self._L.czk_domain_constants(self._h, C.c_uint(log_d), _ptr(out)))
        names = ["size_inv", "group_gen", "group_gen_inv", "generator", "generator_inv", "vanishing_inv"]
        return dict(zip(names, out))

    # ---- pointwise ------------------------------------------------------------------------------
    def fr_vec_op(self, op, a, b, out=None, n=None, mem=CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
            n = a.size // 4
            out = np.empty_like(a) if out is None else out
        self._ck(self._L.czk_fr_vec_op(self._h, C.c_int(op), _ptr(a), _ptr(b), _ptr(out), C.c_size_t(n), C.c_int(mem)))
        return out

    def fr_vec_scale(self, a, k, out=None, n=None, mem=CZK_MEM_HOST):
        if mem == (CZK_MEM_DEVICE | CZK_MEM_SCALAR_HOST):
            k = np.ascontiguousarray(k, np.uint64).reshape(4)
        if mem == CZK_MEM_HOST:
            a, k = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(k, np.uint64)
            n = a.size // 4
            out = np.empty_like(a) if out is None else out
        self._ck(self._L.czk_fr_vec_scale(self._h, _ptr(a), _ptr(k), _ptr(out), C.c_size_t(n), C.c_int(mem)))
        return out

    def fr_powers(self, g, n: int, c=None, out=None, mem=CZK_MEM_HOST):
        """out[i] = c * g^i (g, c: (4,) uint64 Montgomery)."""
        g = np.ascontiguousarray(g, np.uint64).reshape(4)
        cc = None if c is None else np.ascontiguousarray(c, np.uint64).reshape(4)
        if mem == CZK_MEM_HOST:
            out = np.zeros((n, 4), dtype=np.uint64)
        self._ck(self._L.czk_fr_powers(self._h, _ptr(g), _ptr(cc), C.c_size_t(n), _ptr(out), C.c_int(mem)))
        return out

    def fr_beaver_combine(self, x, y, z, sx, oy, add_open, out=None, n=None, mem=CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            x, y, z, sx, oy = (np.ascontiguousarray(v, np.uint64) for v in (x, y, z, sx, oy))
            n = x.size // 4
            out = np.empty_like(x) if out is None else out
        self._ck(self._L.czk_fr_beaver_combine(self._h, _ptr(x), _ptr(y), _ptr(z), _ptr(sx), _ptr(oy), C.c_int(int(add_open)),
                                             _ptr(out), C.c_size_t(n), C.c_int(mem)))
        return out

    def fr_into_repr(self, a, out=None, n=None, mem=CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            a = np.ascontiguousarray(a, np.uint64)
            n = a.size // 4
            out = np.empty_like(a) if out is None else out
        self._ck(self._L.czk_fr_into_repr(self._h, _ptr(a), _ptr(out), C.c_size_t(n), C.c_int(mem)))
        return out

    def fr_from_repr(self, a, out=None, n=None, mem=CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            a = np.ascontiguousarray(a, np.uint64)
            n = a.size // 4
            out = np.empty_like(a) if out is None else out
        self._ck(self._L.czk_fr_from_repr(self._h, _ptr(a), _ptr(out), C.c_size_t(n), C.c_int(mem)))
        return out

    def fr_vec_serialize(self, a, n=None, mem=CZK_MEM_HOST) -> bytes:
        """`Vec<Fr>::serialize`: u64 length prefix + 32 little-endian bytes of into_repr() per element"""
        if mem == CZK_MEM_HOST:
            a = np.ascontiguousarray(a, np.uint64)
            n = a.size // 4
        out = np.zeros(8 + 32 * n, dtype=np.uint8)
        self._ck(self._L.czk_fr_vec_serialize(self._h, _ptr(a if n else None), C.c_size_t(n), C.c_int(mem), _ptr(out)))
        return out.tobytes()

    def fr_vec_deserialize(self, data: bytes, out=None, cap=None, mem=CZK_MEM_HOST):
        """inverse of fr_vec_serialize; host mode returns (n, 4) Montgomery limbs, device mode the element count"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        if mem == CZK_MEM_HOST:
            cap = max(0, (len(data) - 8) // 32)
            out = np.zeros((cap, 4), dtype=np.uint64)
        n = C.c_size_t(0)
        self._ck(self._L.czk_fr_vec_deserialize(self._h, _ptr(buf), C.c_size_t(len(data)), _ptr(out if cap else None), C.c_size_t(cap), C.c_int(mem), C.byref(n)))
        return out[: n.value] if mem == CZK_MEM_HOST else n.value

    # ---- MSM ------------------------------------------------------------------------------------
    def register_bases(self, group: int, bases, inf=None, n: int | None = None, mem: int = CZK_MEM_HOST) -> "Bases":
        aw = 12 if group == CZK_G1 else 24
        if (mem & ~(CZK_MEM_NO_TABLES | CZK_MEM_ANY_POINTS | CZK_MEM_CHECK_SUBGROUP)) == CZK_MEM_HOST:
            bases = np.ascontiguousarray(bases, np.uint64)
            n = bases.size // aw
            if inf is not None:
                inf = np.ascontiguousarray(inf, np.uint8)
        h = C.c_void_p(0)
        self._ck(self._L.czk_bases_register(self._h, C.c_int(group), _ptr(bases), _ptr(inf), C.c_size_t(n), C.c_int(mem), C.byref(h)))
        return Bases(self, h, group, n)

    def msm(self, bases: "Bases", scalars, n_scalars: int | None = None, lanes: int = 1, scalar_form: int = CZK_SCALAR_CANONICAL,
            mem: int = CZK_MEM_HOST):
        """VariableBaseMSM::multi_scalar_mul over `lanes` scalar vectors; returns (lanes, 18|36) Jacobian limbs."""
        jw = 18 if bases.group == CZK_G1 else 36
        if mem == CZK_MEM_HOST:
            scalars = np.ascontiguousarray(scalars, np.uint64)
            if n_scalars is None:
                n_scalars = scalars.size // (4 * lanes)
        out = np.zeros((lanes, jw), dtype=np.uint64)
        self._ck(self._L.czk_msm(self._h, bases._h, _ptr(scalars), C.c_size_t(n_scalars), C.c_size_t(lanes), C.c_int(scalar_form),
                               C.c_int(mem), _ptr(out)))
        return out

    def msm_async(self, bases: "Bases", scalars_ptr, n_scalars: int, lanes: int, scalar_form: int, out: np.ndarray, stable: bool = False,
                  same_scalars: bool = False):
        """czk_msm_async on device scalars; `out` (numpy, lanes x 18|36) is valid after sync().  stable=True promises
        the scalars stay untouched until then (CZK_MEM_STABLE); same_scalars=True (with stable) says they are the previous
        czk_msm_async call's scalars, whose digit sort the library may then take over (CZK_MEM_SAME_SCALARS)."""
        self._ck(self._L.czk_msm_async(self._h, bases._h, _ptr(scalars_ptr), C.c_size_t(n_scalars), C.c_size_t(lanes), C.c_int(scalar_form),
                                     C.c_int(CZK_MEM_DEVICE | (16 if stable else 0) | (CZK_MEM_SAME_SCALARS if same_scalars else 0)), _ptr(out)))
        return out

    def msm_oneshot(self, group, bases, inf, scalars, lanes=1, scalar_form=CZK_SCALAR_CANONICAL):
        aw, jw = (12, 18) if group == CZK_G1 else (24, 36)
        bases = np.ascontiguousarray(bases, np.uint64)
        scalars = np.ascontiguousarray(scalars, np.uint64)
        n = bases.size // aw
        inf = None if inf is None else np.ascontiguousarray(inf, np.uint8)
        out = np.zeros((lanes, jw), dtype=np.uint64)
        fn = self._L.czk_msm_g1 if group == CZK_G1 else self._L.czk_msm_g2
        self._ck(fn(self._h, _ptr(bases), _ptr(inf), _ptr(scalars), C.c_size_t(n), C.c_size_t(lanes), C.c_int(scalar_form), _ptr(out)))
        return out

    def jac_to_affine(self, group, jac):
        aw, jw = (12, 18) if group == CZK_G1 else (24, 36)
        jac = np.ascontiguousarray(jac, np.uint64).reshape(-1, jw)
        n = jac.shape[0]
        aff = np.zeros((n, aw), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self._ck(self._L.czk_jac_to_affine(self._h, C.c_int(group), _ptr(jac), C.c_size_t(n), _ptr(aff), _ptr(inf)))
        return aff, inf

    def jac_add(self, group, a, b):
        jw = 18 if group == CZK_G1 else 36
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        out = np.zeros(jw, dtype=np.uint64)
        self._ck(self._L.czk_jac_add(self._h, C.c_int(group), _ptr(a), _ptr(b), _ptr(out)))
        return out

    def jac_add_mixed(self, group, a, b_aff, b_inf=False):
        jw = 18 if group == CZK_G1 else 36
        a, b_aff = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b_aff, np.uint64)
        out = np.zeros(jw, dtype=np.uint64)
        self._ck(self._L.czk_jac_add_mixed(self._h, C.c_int(group), _ptr(a), _ptr(b_aff), C.c_int(int(b_inf)), _ptr(out)))
        return out

    def jac_scalar_mul(self, group, a, k, scalar_form=CZK_SCALAR_CANONICAL):
        """ProjectiveCurve::mul on a host Jacobian value; k: (4,) uint64 (canonical, or Montgomery with CZK_SCALAR_MONTGOMERY)"""
        jw = 18 if group == CZK_G1 else 36
        a, k = np.ascontiguousarray(a, np.uint64).reshape(jw), np.ascontiguousarray(k, np.uint64).reshape(4)
        out = np.zeros(jw, dtype=np.uint64)
        self._ck(self._L.czk_jac_scalar_mul(self._h, C.c_int(group), _ptr(a), _ptr(k), C.c_int(scalar_form), _ptr(out)))
        return out

    def jac_neg(self, group, a):
        jw = 18 if group == CZK_G1 else 36
        a = np.ascontiguousarray(a, np.uint64).reshape(jw)
        out = np.zeros(jw, dtype=np.uint64)
        self._ck(self._L.czk_jac_neg(self._h, C.c_int(group), _ptr(a), _ptr(out)))
        return out

    def fr_copy_3d(self, dst_ptr, dst_stride, src_ptr, src_stride, n):
        """czk_fr_copy_3d: strided copy (src_ptr None: zero fill) of Fr elements in device memory, in stream order; strides / extents: 3 ints each"""
        ds = (C.c_size_t * 3)(*[int(v) for v in dst_stride])
        ss = (C.c_size_t * 3)(*[int(v) for v in (src_stride or (0, 0, 0))])
        nn = (C.c_size_t * 3)(*[int(v) for v in n])
        self._ck(self._L.czk_fr_copy_3d(self._h, _ptr(dst_ptr), ds, _ptr(src_ptr), ss if src_ptr is not None else None, nn))

    def fr_spdz_open(self, shares_ptr, parties: int, n: int, out_value_ptr) -> int:
        """Local part of SpdzFieldShare::batch_open on device buffers; returns the number of failed MAC checks."""
        bad = C.c_uint64(0)
        self._ck(self._L.czk_fr_spdz_open(self._h, _ptr(shares_ptr), C.c_size_t(parties), C.c_size_t(n), _ptr(out_value_ptr), C.byref(bad)))
        return bad.value

    def fr_lanes_sum(self, x_ptr, k: int, n: int, out_ptr=None, count_nonzero: bool = False):
        """out[i] = sum of k device vectors; returns the number of non-zero sums when count_nonzero."""
        nz = C.c_uint64(0)
        self._ck(self._L.czk_fr_lanes_sum(self._h, _ptr(x_ptr), C.c_size_t(k), C.c_size_t(n), _ptr(out_ptr), C.byref(nz) if count_nonzero else C.c_void_p(0)))
        return nz.value

    def fr_spdz_dx(self, value_ptr, mac_ptr, mac_share, out_ptr, n: int):
        """dx_t = mac_share * value - mac on device vectors (share/spdz.rs:176-180); mac_share: (4,) uint64 Montgomery."""
        ms = np.ascontiguousarray(mac_share, np.uint64).reshape(4)
        self._ck(self._L.czk_fr_spdz_dx(self._h, _ptr(value_ptr), _ptr(mac_ptr), _ptr(ms), _ptr(out_ptr), C.c_size_t(n)))

    def share_domain_constants(self, parties: int):
        out = np.zeros((3, 4), dtype=np.uint64)
        self._ck(self._L.czk_share_domain_constants(self._h, C.c_size_t(parties), _ptr(out)))
        return dict(zip(["size_inv", "group_gen", "group_gen_inv"], out))

    def fr_gsz_open(self, shares_ptr, parties: int, n: int, out_value_ptr, degree: int = 0, degrees_ptr=None) -> int:
        """Local part of GszFieldShare::batch_open on device buffers; returns the number of degree-bound violations."""
        bad = C.c_uint64(0)
        self._ck(self._L.czk_fr_gsz_open(self._h, _ptr(shares_ptr), C.c_size_t(parties), C.c_size_t(n), _ptr(degrees_ptr), C.c_uint(degree),
                                       _ptr(out_value_ptr), C.byref(bad)))
        return bad.value

    def r1cs_matrix_register(self, row_ptr, col_idx, coeff, n_vars: int, mem=CZK_MEM_HOST, m=None, nnz=None) -> "R1csMatrix":
        """One of ConstraintMatrices::{a, b, c} in CSR form (host numpy arrays, or device pointers with m / nnz given)."""
        if mem == CZK_MEM_HOST:
            row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
            col_idx = np.ascontiguousarray(col_idx, np.uint32)
            coeff = np.ascontiguousarray(coeff, np.uint64)
            m, nnz = row_ptr.size - 1, col_idx.size
        h = C.c_void_p(0)
        self._ck(self._L.czk_r1cs_matrix_register(self._h, _ptr(row_ptr), _ptr(col_idx), _ptr(coeff), C.c_size_t(m), C.c_size_t(nnz),
                                                C.c_size_t(n_vars), C.c_int(mem), C.byref(h)))
        return R1csMatrix(self, h, m, n_vars)

    def r1cs_matvec(self, mat: "R1csMatrix", z, lanes: int = 1, out=None, z_stride=None, out_stride=None, mem=CZK_MEM_HOST):
        """evaluate_constraint over every row and lane; host mode returns (lanes, m, 4)."""
        if mem == CZK_MEM_HOST:
            z = np.ascontiguousarray(z, np.uint64).reshape(lanes, -1, 4)
            z_stride = z.shape[1]
            out_stride = mat.m
            out = np.zeros((lanes, mat.m, 4), dtype=np.uint64)
        self._ck(self._L.czk_r1cs_matvec(self._h, mat._h, _ptr(z), C.c_size_t(z_stride), C.c_size_t(lanes), _ptr(out), C.c_size_t(out_stride), C.c_int(mem)))
        return out

    def poly_div_linear(self, coeffs, z, lanes: int = 1, n=None, quotient=None, remainder=None, mem=CZK_MEM_HOST):
        """coeffs / (X - z) per lane; host mode returns (quotient (lanes, n-1, 4), remainder (lanes, 4))."""
        z = np.ascontiguousarray(z, np.uint64).reshape(4)
        if mem == CZK_MEM_HOST:
            coeffs = np.ascontiguousarray(coeffs, np.uint64).reshape(lanes, -1, 4)
            n = coeffs.shape[1]
            quotient = np.zeros((lanes, max(n - 1, 0), 4), dtype=np.uint64)
            remainder = np.zeros((lanes, 4), dtype=np.uint64)
        qp = quotient if not (isinstance(quotient, np.ndarray) and quotient.size == 0) else None
        cp = coeffs if not (isinstance(coeffs, np.ndarray) and coeffs.size == 0) else None
        self._ck(self._L.czk_poly_div_linear(self._h, _ptr(cp), C.c_size_t(n), C.c_size_t(lanes), _ptr(z), _ptr(qp), _ptr(remainder), C.c_int(mem)))
        return quotient, remainder

    def poly_div_vanishing(self, coeffs, n: int, lanes: int = 1, m=None, quotient=None, remainder=None, mem=CZK_MEM_HOST):
        """coeffs = q (X^n - 1) + r per lane (czk_poly_div_vanishing); host mode returns (q (lanes, m - n, 4), r (lanes, n, 4))."""
        if mem == CZK_MEM_HOST:
            coeffs = np.ascontiguousarray(coeffs, np.uint64).reshape(lanes, -1, 4)
            m = coeffs.shape[1]
            quotient = np.zeros((lanes, max(m - n, 0), 4), dtype=np.uint64)
            remainder = np.zeros((lanes, n, 4), dtype=np.uint64)
        qp = quotient if not (isinstance(quotient, np.ndarray) and quotient.size == 0) else None
        self._ck(self._L.czk_poly_div_vanishing(self._h, _ptr(coeffs), C.c_size_t(m), C.c_size_t(lanes), C.c_size_t(n), _ptr(qp), _ptr(remainder), C.c_int(mem)))
        return quotient, remainder

    def poly_evaluate(self, coeffs, z, lanes: int = 1, n=None, values=None, mem=CZK_MEM_HOST):
        """p(z) per lane (czk_poly_evaluate); host mode returns (lanes, 4)."""
        z = np.ascontiguousarray(z, np.uint64).reshape(4)
        if mem == CZK_MEM_HOST:
            coeffs = np.ascontiguousarray(coeffs, np.uint64).reshape(lanes, -1, 4)
            n = coeffs.shape[1]
            values = np.zeros((lanes, 4), dtype=np.uint64)
        cp = coeffs if not (isinstance(coeffs, np.ndarray) and coeffs.size == 0) else None
        self._ck(self._L.czk_poly_evaluate(self._h, _ptr(cp), C.c_size_t(n), C.c_size_t(lanes), _ptr(z), _ptr(values), C.c_int(mem)))
        return values

    def poly_evaluate_many(self, ptrs, ns, lanes, zs, value_ptrs):
        """czk_poly_evaluate_many on DEVICE pointers: polynomial k = ns[k] coefficients x lanes[k] lanes at ptrs[k], evaluated at zs[k] (Montgomery limbs),
        its values written to value_ptrs[k]."""
        k = len(ptrs)
        src = (C.c_void_p * k)(*[int(p) for p in ptrs])
        dst = (C.c_void_p * k)(*[int(p) for p in value_ptrs])
        n = (C.c_size_t * k)(*[int(x) for x in ns])
        ln = (C.c_size_t * k)(*[int(x) for x in lanes])
        z = np.ascontiguousarray(np.stack([np.asarray(x, np.uint64).reshape(4) for x in zs]) if k else np.zeros((0, 4), np.uint64))
        self._ck(self._L.czk_poly_evaluate_many(self._h, C.c_size_t(k), src, n, ln, _ptr(z if k else None), dst))

    def fr_lincomb(self, ptrs, lens, term_lanes, coeffs, lanes: int, lift_mask: int, out_ptr, out_len: int, constant=None):
        """czk_fr_lincomb on DEVICE pointers: out[l][i] = constant + sum_k coeffs[k] * term_k[l][i]; a term with one lane -- and the constant -- is public
        (added on the lanes of lift_mask)."""
        k = len(ptrs)
        src = (C.c_void_p * k)(*[int(p) for p in ptrs])
        n = (C.c_size_t * k)(*[int(x) for x in lens])
        tl = (C.c_size_t * k)(*[int(x) for x in term_lanes])
        c = np.ascontiguousarray(np.stack([np.asarray(x, np.uint64).reshape(4) for x in coeffs]) if k else np.zeros((0, 4), np.uint64))
        cst = None if constant is None else np.ascontiguousarray(constant, np.uint64).reshape(4)
        self._ck(self._L.czk_fr_lincomb(self._h, C.c_size_t(k), src, n, tl, _ptr(c if k else None), _ptr(cst), C.c_size_t(lanes), C.c_uint64(lift_mask), _ptr(out_ptr), C.c_size_t(out_len)))

    def fr_prefix_product(self, x, out=None, n=None, mem=CZK_MEM_HOST):
        """Running products of a public Fr vector (partial_products' local loop)."""
        if mem == CZK_MEM_HOST:
            x = np.ascontiguousarray(x, np.uint64).reshape(-1, 4)
            n = x.shape[0]
            out = np.zeros_like(x)
        self._ck(self._L.czk_fr_prefix_product(self._h, _ptr(x if n else None), C.c_size_t(n), _ptr(out if n else None), C.c_int(mem)))
        return out

    def fr_batch_inverse(self, v, coeff=None, out=None, n=None, mem=CZK_MEM_HOST):
        """batch_inversion_and_mul: out[i] = coeff / v[i] (zeros stay zero)."""
        if mem == CZK_MEM_HOST:
            v = np.ascontiguousarray(v, np.uint64).reshape(-1, 4)
            n = v.shape[0]
            out = np.zeros_like(v)
        if coeff is not None:
            coeff = np.ascontiguousarray(coeff, np.uint64).reshape(4)
        self._ck(self._L.czk_fr_batch_inverse(self._h, _ptr(v if n else None), C.c_size_t(n), _ptr(coeff), _ptr(out if n else None), C.c_int(mem)))
        return out

    def fixed_base_points(self, group, k, out=None, n=None, mem=CZK_MEM_HOST):
        aw = 12 if group == CZK_G1 else 24
        if mem == CZK_MEM_HOST:
            k = np.ascontiguousarray(k, np.uint64)
            n = k.size // 4
            out = np.zeros((n, aw), dtype=np.uint64)
        self._ck(self._L.czk_fixed_base_points(self._h, C.c_int(group), _ptr(k), C.c_size_t(n), _ptr(out), C.c_int(mem)))
        return out

    # ---- measurement hooks ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._ck(self._L.czk_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile_reset(self):
        self._ck(self._L.czk_profile_reset(self._h))

    def profile_read(self, kernel: str):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._ck(self._L.czk_profile_read(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_intervals(self, kernel: str):
        """(start, stop) in ms after this context's profile origin of every bracket of `kernel`: numpy array (n, 2)"""
        n = C.c_size_t(0)
        self._ck(self._L.czk_profile_intervals(self._h, kernel.encode(), None, None, C.c_size_t(0), C.byref(n)))
        a, b = np.zeros(n.value, np.float64), np.zeros(n.value, np.float64)
        if n.value:
            self._ck(self._L.czk_profile_intervals(self._h, kernel.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                                 C.c_size_t(n.value), C.byref(n)))
        return np.stack([a, b], axis=1)

    def profile_base_offset(self, other) -> float:
        """other's profile origin minus this context's, in ms (both after profile_reset)"""
        ms = C.c_double(0)
        self._ck(self._L.czk_profile_base_offset(self._h, other._h, C.byref(ms)))
        return ms.value

    # ---- Groth16 witness map (device buffers) ------------------------------------------------------
    def witness_map_pre(self, a_ptr, b_ptr, log_d, lanes, a_len=None, b_len=None):
        """a_len / b_len: evaluations present in each lane (default D); the rest of the domain counts as zero."""
        d = 1 << log_d
        self._ck(self._L.czk_witness_map_pre(self._h, _ptr(a_ptr), C.c_size_t(d if a_len is None else a_len), _ptr(b_ptr),
                                           C.c_size_t(d if b_len is None else b_len), C.c_uint(log_d), C.c_size_t(lanes)))

    def witness_map_post(self, ab_ptr, c_ptr, log_d, lanes, c_len=None):
        self._ck(self._L.czk_witness_map_post(self._h, _ptr(ab_ptr), _ptr(c_ptr), C.c_size_t((1 << log_d) if c_len is None else c_len),
                                            C.c_uint(log_d), C.c_size_t(lanes)))


class Net:
    """czk_net: mpc-net between czk contexts (include/czk.h): RCCL for one party per GPU, SHM for parties that are processes of one node in
    any assignment to GPUs.  `ctx` may be None for the SHM transport (host-memory primitives only).  Device buffers are raw pointers."""

    STATS = ("bytes_sent", "bytes_recv", "broadcasts", "to_king", "from_king")

    def __init__(self, ctx, transport: int, rank: int, world: int, id_bytes: bytes, options: dict | None = None, lab: bool = False):
        self.ctx = ctx
        self._L = ctx._L if ctx is not None else (lab_lib() if lab else lib())
        self._h = C.c_void_p(0)
        self.rank, self.world, self.transport = rank, world, transport
        idb = (C.c_uint8 * len(id_bytes)).from_buffer_copy(bytes(id_bytes))
        rc = self._L.czk_net_create(ctx._h if ctx is not None else None, C.c_int(transport), C.c_int(rank), C.c_int(world), idb, C.c_size_t(len(id_bytes)),
                                    C.byref(self._h))
        if rc:
            raise CzkError(rc, (self._L.czk_last_error(ctx._h) or b"").decode() if ctx is not None else "czk_net_create failed")
        for k, v in (options or {}).items():
            self.set_option(k, v)

    @staticmethod
    def unique_id(transport: int, lab: bool = False) -> bytes:
        """czk_net_unique_id: rank 0 calls this and hands the bytes to the other ranks (RCCL: ncclGetUniqueId; SHM: 16 random bytes)"""
        L = lab_lib() if lab else lib()
        buf = (C.c_uint8 * 128)()
        n = C.c_size_t(0)
        rc = L.czk_net_unique_id(C.c_int(transport), buf, C.c_size_t(128), C.byref(n))
        if rc:
            raise CzkError(rc, "czk_net_unique_id failed (RCCL: librccl.so.1 not loadable)")
        return bytes(buf[: n.value])

    def _ck(self, rc):
        if rc:
            raise CzkError(rc, (self._L.czk_net_last_error(self._h) or b"").decode())

    def set_option(self, name: str, value: int):
        self._ck(self._L.czk_net_set_option(self._h, name.encode(), C.c_long(int(value))))

    def stats(self) -> dict:
        out = (C.c_uint64 * 5)()
        self._ck(self._L.czk_net_stats(self._h, out))
        return dict(zip(self.STATS, [int(v) for v in out]))

    def stats_reset(self):
        self._L.czk_net_stats_reset(self._h)

    def barrier(self):
        self._ck(self._L.czk_net_barrier(self._h))

    # ---- byte primitives: host numpy uint8 arrays (returned), or device pointers with `nbytes` given -------------------------
    def broadcast(self, send, nbytes: int | None = None, recv=None, mem: int = CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            send = np.ascontiguousarray(send).view(np.uint8).reshape(-1)
            nbytes = send.size
            recv = np.zeros((self.world, nbytes), dtype=np.uint8)
        self._ck(self._L.czk_net_broadcast(self._h, _ptr(send if nbytes else None), C.c_size_t(nbytes), _ptr(recv if nbytes else None), C.c_int(mem)))
        return recv

    def send_to_king(self, send, nbytes: int | None = None, recv=None, mem: int = CZK_MEM_HOST):
        if mem == CZK_MEM_HOST:
            send = np.ascontiguousarray(send).view(np.uint8).reshape(-1)
            nbytes = send.size
            recv = np.zeros((self.world, nbytes), dtype=np.uint8) if self.rank == 0 else None
        self._ck(self._L.czk_net_send_to_king(self._h, _ptr(send if nbytes else None), C.c_size_t(nbytes), _ptr(recv if nbytes else None), C.c_int(mem)))
        return recv

    def recv_from_king(self, send, nbytes: int | None = None, recv=None, mem: int = CZK_MEM_HOST):
        """send: (world, nbytes) on the king, None elsewhere (host mode: nbytes must then be given)"""
        if mem == CZK_MEM_HOST:
            if send is not None:
                send = np.ascontiguousarray(send).view(np.uint8).reshape(self.world, -1)
                nbytes = send.shape[1]
            recv = np.zeros(nbytes, dtype=np.uint8)
        self._ck(self._L.czk_net_recv_from_king(self._h, _ptr(send if nbytes else None), C.c_size_t(nbytes), _ptr(recv if nbytes else None), C.c_int(mem)))
        return recv

    def atomic_broadcast(self, x_ptr, n: int, recv_ptr, rand32: bytes | None = None, mem: int = CZK_MEM_DEVICE):
        r = (C.c_uint8 * 32).from_buffer_copy(rand32) if rand32 is not None else None
        self._ck(self._L.czk_net_atomic_broadcast(self._h, _ptr(x_ptr), C.c_size_t(n), _ptr(recv_ptr), r, C.c_int(mem)))

    # ---- the reference's batch opens on device lanes ---------------------------------------------------------------------------
    def spdz_batch_open(self, sh_ptr, mac_ptr, mac_share, n: int, out_ptr, commit: bool = True) -> int:
        """SpdzFieldShare::batch_open; returns the number of failed MAC checks (the reference asserts 0).  commit=True (the default, as in the
        reference: spdz.rs:179 -> channel.rs:50-75) sends dx_ts through the commit-then-open round; False is an explicit opt-out."""
        ms = np.ascontiguousarray(mac_share, np.uint64).reshape(4)
        bad = C.c_uint64(0)
        self._ck(self._L.czk_spdz_batch_open(self._h, _ptr(sh_ptr), _ptr(mac_ptr), _ptr(ms), C.c_size_t(n), _ptr(out_ptr),
                                             C.c_int(CZK_OPEN_COMMIT if commit else 0), C.byref(bad)))
        return bad.value

    def add_batch_open(self, val_ptr, n: int, out_ptr):
        self._ck(self._L.czk_add_batch_open(self._h, _ptr(val_ptr), C.c_size_t(n), _ptr(out_ptr)))

    def gsz_batch_open(self, val_ptr, n: int, out_ptr, degree: int = 0, degrees_ptr=None) -> int:
        bad = C.c_uint64(0)
        self._ck(self._L.czk_gsz_batch_open(self._h, _ptr(val_ptr), C.c_size_t(n), _ptr(degrees_ptr), C.c_uint(degree), _ptr(out_ptr), C.byref(bad)))
        return bad.value

    def gsz_batch_king_compute(self, val_ptr, n: int, out_ptr, degree: int = 0, degrees_ptr=None) -> int:
        bad = C.c_uint64(0)
        self._ck(self._L.czk_gsz_batch_king_compute(self._h, _ptr(val_ptr), C.c_size_t(n), _ptr(degrees_ptr), C.c_uint(degree), _ptr(out_ptr), C.byref(bad)))
        return bad.value

    def fr_send_to_king(self, x_ptr, n: int, gathered_ptr):
        self._ck(self._L.czk_fr_send_to_king(self._h, _ptr(x_ptr), C.c_size_t(n), _ptr(gathered_ptr)))

    def fr_recv_from_king(self, parts_ptr, n: int, out_ptr):
        self._ck(self._L.czk_fr_recv_from_king(self._h, _ptr(parts_ptr), C.c_size_t(n), _ptr(out_ptr)))

    def close(self):
        if self._h:
            self._L.czk_net_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sha256(data: bytes, lab: bool = False) -> bytes:
    """czk_sha256 (the library's CommitHash): for tests against hashlib"""
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    (lab_lib() if lab else lib()).czk_sha256(buf, C.c_size_t(len(data)), out)
    return bytes(out)


class Lanes:
    """czk_lanes: share lanes resident on the GPU (lane-major, 4 u64 per element)."""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self._h = ctx, handle
        self.lanes, self.len = int(self.ctx._L.czk_lanes_count(handle)), int(self.ctx._L.czk_lanes_len(handle))

    def ptr(self, lane: int = 0, elem: int = 0) -> int:
        """Device address of one element (0 when outside the allocation): use with CZK_MEM_DEVICE."""
        return self.ctx._L.czk_lanes_data(self._h, C.c_size_t(lane), C.c_size_t(elem)) or 0

    def upload(self, host, lane: int = 0, elem: int = 0):
        host = np.ascontiguousarray(host, np.uint64)
        self.ctx._ck(self.ctx._L.czk_lanes_upload(self.ctx._h, self._h, C.c_size_t(lane), C.c_size_t(elem), _ptr(host if host.size else None), C.c_size_t(host.size // 4)))

    def download(self, lane: int = 0, elem: int = 0, n: int | None = None) -> np.ndarray:
        n = self.len - elem if n is None else n
        out = np.zeros((n, 4), dtype=np.uint64)
        self.ctx._ck(self.ctx._L.czk_lanes_download(self.ctx._h, self._h, C.c_size_t(lane), C.c_size_t(elem), _ptr(out if n else None), C.c_size_t(n)))
        return out

    def download_deferred(self, out: np.ndarray, lane: int = 0, elem: int = 0):
        """czk_lanes_download_deferred: `out` (n x 4 u64, kept alive by the caller) is filled when a later mark is waited for / at the next sync"""
        assert out.dtype == np.uint64 and out.flags.c_contiguous
        n = out.size // 4
        self.ctx._ck(self.ctx._L.czk_lanes_download_deferred(self.ctx._h, self._h, C.c_size_t(lane), C.c_size_t(elem), _ptr(out if n else None), C.c_size_t(n)))

    def copy_from(self, src: "Lanes", n: int, lane: int = 0, elem: int = 0, src_lane: int = 0, src_elem: int = 0):
        self.ctx._ck(self.ctx._L.czk_lanes_copy(self.ctx._h, self._h, C.c_size_t(lane), C.c_size_t(elem), src._h, C.c_size_t(src_lane), C.c_size_t(src_elem), C.c_size_t(n)))

    def zero(self, n: int, lane: int = 0, elem: int = 0):
        self.ctx._ck(self.ctx._L.czk_lanes_zero(self.ctx._h, self._h, C.c_size_t(lane), C.c_size_t(elem), C.c_size_t(n)))

    def free(self):
        if self._h:
            self.ctx._L.czk_lanes_free(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class R1csMatrix:
    """czk_r1cs_matrix: a public constraint matrix pinned in HBM (CSR)."""

    def __init__(self, ctx: "Context", handle, m: int, n_vars: int):
        self.ctx, self._h, self.m, self.n_vars = ctx, handle, m, n_vars

    def release(self):
        if self._h:
            self.ctx._L.czk_r1cs_matrix_release(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Bases:
    """czk_bases: a public base array pinned (with its window multiples) in HBM."""

    def __init__(self, ctx: Context, handle, group: int, n: int):
        self.ctx, self._h, self.group, self.n = ctx, handle, group, n

    def __len__(self):
        return int(self.ctx._L.czk_bases_len(self._h))

    def layout(self):
        """(window width c, number of windows) chosen at registration."""
        c, w = C.c_uint(0), C.c_uint(0)
        self.ctx._L.czk_bases_layout(self._h, C.byref(c), C.byref(w))
        return c.value, w.value

    def windows(self) -> int:
        return self.layout()[1]

    def arith(self) -> int:
        """0 = XYZZ saturated, 1 = XYZZ unsaturated, 2 = twisted Edwards (czk_bases_arith)"""
        return int(self.ctx._L.czk_bases_arith(self._h))

    def check_subgroup(self) -> int:
        """Number of registered bases outside the prime-order subgroup (czk_bases_check_subgroup: [r] P == infinity on the GPU)."""
        bad = C.c_size_t(0)
        self.ctx._ck(self.ctx._L.czk_bases_check_subgroup(self.ctx._h, self._h, C.byref(bad)))
        return int(bad.value)

    def layout_for(self, n_scalars: int):
        """(c, windows) an MSM of n_scalars scalars over these bases runs with (czk_bases_layout_for)."""
        c, w = C.c_uint(0), C.c_uint(0)
        self.ctx._L.czk_bases_layout_for(self._h, C.c_size_t(n_scalars), C.byref(c), C.byref(w))
        return c.value, w.value

    def prepare(self, n_scalars: int):
        """Builds the table set MSMs of n_scalars scalars will use, up front (czk_bases_prepare)."""
        self.ctx._ck(self.ctx._L.czk_bases_prepare(self.ctx._h, self._h, C.c_size_t(n_scalars)))

    def release(self):
        if self._h:
            self.ctx._L.czk_bases_release(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

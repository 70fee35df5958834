"""A polyvm backend that tracks SHAPES only and records every hot-path call a prover makes, in order: the NTTs (domain size,
kind, lanes, input length), the commitments (MSM length, lanes) and the divisions / scans / inversions.  No arithmetic, no GPU.

Used to audit collaborative-zksnark_amd/polyvm.py against a call-site list written by hand from the reference's source
(tests/golden/plonk_marlin_callsites.json; tests/test_callsite_audit.py) and to size the MSM classes."""
import numpy as np

from czk_amd.polyvm import Backend, Pending


class Arr:
    __slots__ = ("lanes", "n")

    def __init__(self, lanes, n):
        self.lanes, self.n = int(lanes), int(n)

    def __repr__(self):
        return f"Arr({self.lanes}, {self.n})"


KINDS = {0: "fft", 1: "ifft", 2: "coset_fft", 3: "coset_ifft"}


class ShapeBackend(Backend):
    def __init__(self, lanes, lift=None):
        self.lanes = lanes
        self.lift = tuple([1] * lanes) if lift is None else tuple(lift)
        self.log = []          # (op, size, lanes[, extra])

    # --- storage
    def zeros(self, lanes, n): return Arr(lanes, n)
    def lanes_of(self, a): return a.lanes
    def lane_stack(self, parts): return Arr(len(parts), parts[0].n)
    def upload(self, a):
        a = np.asarray(a)
        return Arr(1, a.shape[0]) if a.ndim == 2 else Arr(a.shape[0], a.shape[1])
    def download(self, a): return np.zeros((a.lanes, a.n, 4), dtype=np.uint64)
    def length(self, a): return a.n
    def resized(self, a, n): return Arr(a.lanes, n)
    def drop_first(self, a, k): return Arr(a.lanes, max(a.n - k, 0))
    def concat(self, parts): return Arr(parts[0].lanes, sum(p.n for p in parts))
    def strided_split(self, a, n): return Arr(a.lanes * n, a.n // n)
    def strided_merge(self, a, n, lanes): return Arr(lanes, a.n * n)
    def const(self, k, n): return Arr(1, n)

    # --- transforms and arithmetic
    def ntt(self, a, size, kind):
        self.log.append(("ntt", int(size), a.lanes, KINDS[kind], min(a.n, int(size))))
        return Arr(a.lanes, size)

    def _same(self, a, b):
        assert a.lanes == b.lanes and a.n == b.n, (a, b)
        return Arr(a.lanes, a.n)

    def add(self, a, b): return self._same(a, b)
    def sub(self, a, b): return self._same(a, b)

    def mul(self, a, b):
        assert a.n == b.n and (a.lanes == b.lanes or 1 in (a.lanes, b.lanes)), (a, b)
        return Arr(max(a.lanes, b.lanes), a.n)

    def scale(self, a, k): return Arr(a.lanes, a.n)
    def powers(self, g, n): return Arr(1, n)

    def div_linear(self, a, z):
        self.log.append(("div_linear", a.n, a.lanes))
        return Arr(a.lanes, max(a.n - 1, 0)), np.zeros((a.lanes, 4), dtype=np.uint64)

    def evaluate(self, a, x, public=None):
        """Polynomial::evaluate at one point (the GPU backend runs it as the remainder of a division by X - x)"""
        self.log.append(("evaluate", a.n, a.lanes))
        return np.zeros((a.lanes, 4), dtype=np.uint64)

    def div_vanishing(self, a, n):
        """divide_by_vanishing_poly in coefficient form: an O(len) pass of additions in the reference; logged as one operation"""
        self.log.append(("div_vanishing", a.n, a.lanes, int(n)))
        return Arr(a.lanes, max(a.n - n, 0)), Arr(a.lanes, min(a.n, n))

    def prefix_product(self, a):
        self.log.append(("prefix_product", a.n, a.lanes))
        return Arr(a.lanes, a.n)

    def inverse(self, a):
        self.log.append(("batch_inverse", a.n, a.lanes))
        return Arr(a.lanes, a.n)

    def jac_add_mixed(self, a_jac, b_aff, b_inf): return np.zeros(18, dtype=np.uint64)
    def jac_to_affine(self, jac): return np.zeros((len(jac), 12), dtype=np.uint64), np.zeros(len(jac), dtype=np.uint8)

    def commit(self, a, key="g"):
        self.log.append(("msm" if key == "g" else "msm_gamma", a.n, a.lanes))
        p = Pending(None)
        p.value = (np.zeros((a.lanes, 12), dtype=np.uint64), np.zeros(a.lanes, dtype=np.uint8))
        return p

    def random(self, seed, n): return Arr(1, n)
    def root_of_unity(self, size): return 7

    def transcript_point(self):
        self.log.append(("transcript_point",))

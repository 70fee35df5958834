"""The polynomial machine of collaborative-zksnark_amd/polyvm.py on top of the CPU checker (oracle/): the second backend the
Plonk / Marlin pipeline tests compare the GPU path with.  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from util import rand_fr_canonical


def make_backend(orc, polyvm, lanes, max_degree, base_seed=0xBA5E5 + 77, lift=None, bases=None, bases_gamma=None):
    class OracleBackend(polyvm.Backend):
        def __init__(self):
            self.lanes = lanes
            self.lift = tuple([1] * lanes) if lift is None else tuple(lift)
            self.bases = bases          # (n, 12) affine Montgomery limbs of [k_i] G, produced once by the caller
            self.bases_gamma = bases_gamma   # powers_of_gamma_g (8 points), likewise

        def upload(self, a):
            a = np.ascontiguousarray(a, dtype=np.uint64)
            return a[None].copy() if a.ndim == 2 else a.copy()

        def download(self, a):
            return a

        def length(self, a):
            return a.shape[1]

        def lanes_of(self, a):
            return a.shape[0]

        def zeros(self, l, n):
            return np.zeros((l, n, 4), dtype=np.uint64)

        def lane_stack(self, parts):
            return np.concatenate(parts, axis=0)

        def concat(self, parts):
            return np.concatenate(parts, axis=1)

        def strided_split(self, a, n):
            lanes, total = a.shape[0], a.shape[1]
            return np.ascontiguousarray(a.reshape(lanes, total // n, n, 4).transpose(0, 2, 1, 3).reshape(lanes * n, total // n, 4))

        def strided_merge(self, a, n, l):
            L = a.shape[1]
            return np.ascontiguousarray(a.reshape(l, n, L, 4).transpose(0, 2, 1, 3).reshape(l, L * n, 4))

        def resized(self, a, n):
            out = np.zeros((a.shape[0], n, 4), dtype=np.uint64)
            m = min(n, a.shape[1])
            out[:, :m] = a[:, :m]
            return out

        def drop_first(self, a, k):
            return a[:, k:].copy()

        def _lanewise(self, f, a, b):
            if a.shape[0] != b.shape[0]:
                if a.shape[0] == 1:
                    a = np.broadcast_to(a, b.shape)
                else:
                    b = np.broadcast_to(b, a.shape)
            assert a.shape == b.shape, (a.shape, b.shape)
            return np.stack([f(np.ascontiguousarray(a[l]), np.ascontiguousarray(b[l])) for l in range(a.shape[0])])

        def add(self, a, b):
            assert a.shape == b.shape
            return self._lanewise(orc.fr_add, a, b)

        def sub(self, a, b):
            assert a.shape == b.shape
            return self._lanewise(orc.fr_sub, a, b)

        def mul(self, a, b):
            return self._lanewise(orc.fr_mul, a, b)

        def scale(self, a, k):
            kk = np.tile(polyvm.mont(k), (a.shape[1], 1))
            return np.stack([orc.fr_mul(np.ascontiguousarray(a[l]), kk) for l in range(a.shape[0])])

        def powers(self, g, n):
            out = np.zeros((n, 4), dtype=np.uint64)
            v = 1
            vals = []
            for _ in range(n):
                vals.append(v)
                v = v * g % polyvm.R_MOD
            out[:] = np.stack([polyvm.mont(x) for x in vals]) if n else out
            return out[None]

        def ntt(self, a, size, kind):
            n = min(a.shape[1], size)
            return np.stack([orc.ntt_fr_mixed(np.ascontiguousarray(a[l, :n]), size, kind, n) for l in range(a.shape[0])])

        def div_linear(self, a, z):
            qs, rs = [], []
            for l in range(a.shape[0]):
                q, r = orc.poly_div_linear(np.ascontiguousarray(a[l]), polyvm.mont(z))
                qs.append(q.reshape(-1, 4))
                rs.append(r.reshape(4))
            return np.stack(qs), np.stack(rs)

        def prefix_product(self, a):
            return np.stack([orc.fr_prefix_product(np.ascontiguousarray(a[l])) for l in range(a.shape[0])])

        def inverse(self, a):
            return np.stack([orc.fr_batch_inverse(np.ascontiguousarray(a[l]), polyvm.mont(1)) for l in range(a.shape[0])])

        def random(self, seed, n):
            return orc.fr_from_repr(rand_fr_canonical(seed, n))[None]

        def root_of_unity(self, size):
            return polyvm.unmont(orc.fr_root_of_unity_mixed(size))

        def jac_add_mixed(self, a_jac, b_aff, b_inf):
            return orc.jac_add_mixed(1, a_jac, b_aff, b_inf)

        def jac_to_affine(self, jac):
            affs, infs = zip(*(orc.jac_to_affine(1, j) for j in jac))
            return np.stack(affs), np.array([1 if i else 0 for i in infs], dtype=np.uint8)

        def commit(self, a, key="g"):
            n = a.shape[1]
            inf = np.zeros(n, dtype=np.uint8)
            bases = self.bases if key == "g" else self.bases_gamma
            affs, infs = [], []
            for l in range(a.shape[0]):
                jac = orc.multi_scalar_mul(1, bases[:n], inf, np.ascontiguousarray(a[l]))
                aff, is_inf = orc.jac_to_affine(1, jac)
                affs.append(aff)
                infs.append(1 if is_inf else 0)
            return np.stack(affs), np.array(infs, dtype=np.uint8)
    return OracleBackend()


def make_lockstep(polyvm, gpu, cpu):
    """Runs every primitive on both backends and compares the results immediately: a parity failure names the first operation
    that diverges.  Arrays are (gpu_array, cpu_array) pairs."""
    class Lockstep(polyvm.Backend):
        lanes, lift = gpu.lanes, gpu.lift

        def __init__(self):
            self.ops = 0

        def _chk(self, name, g, c):
            self.ops += 1
            got = gpu.download(g)
            assert got.shape == c.shape, (name, got.shape, c.shape)
            if not np.array_equal(got, c):
                bad = np.argwhere((got != c).any(axis=2))
                raise AssertionError(f"lockstep: operation #{self.ops} `{name}` diverges at (lane, index) {bad[:4].tolist()} of shape {c.shape}")
            return (g, c)

        def _both(self, name, *args, **kw):
            ga = [a[0] if isinstance(a, tuple) else a for a in args]
            ca = [a[1] if isinstance(a, tuple) else a for a in args]
            return self._chk(name, getattr(gpu, name)(*ga, **kw), getattr(cpu, name)(*ca, **kw))

        def upload(self, a): return self._both("upload", a)
        def download(self, a): return a[1]
        def length(self, a): return cpu.length(a[1])
        def lanes_of(self, a): return cpu.lanes_of(a[1])
        def zeros(self, l, n): return self._both("zeros", l, n)
        def lane_stack(self, parts): return self._chk("lane_stack", gpu.lane_stack([p[0] for p in parts]), cpu.lane_stack([p[1] for p in parts]))
        def concat(self, parts): return self._chk("concat", gpu.concat([p[0] for p in parts]), cpu.concat([p[1] for p in parts]))
        def resized(self, a, n): return self._both("resized", a, n)
        def strided_split(self, a, n): return self._both("strided_split", a, n)
        def strided_merge(self, a, n, l): return self._both("strided_merge", a, n, l)
        def drop_first(self, a, k): return self._both("drop_first", a, k)
        def ntt(self, a, size, kind): return self._both("ntt", a, size, kind)
        def add(self, a, b): return self._both("add", a, b)
        def sub(self, a, b): return self._both("sub", a, b)
        def mul(self, a, b): return self._both("mul", a, b)
        def scale(self, a, k): return self._both("scale", a, k)
        def powers(self, g, n): return self._both("powers", g, n)
        def prefix_product(self, a): return self._both("prefix_product", a)
        def inverse(self, a): return self._both("inverse", a)
        def random(self, seed, n): return self._both("random", seed, n)
        def root_of_unity(self, size):
            a, b = gpu.root_of_unity(size), cpu.root_of_unity(size)
            assert a == b, "root_of_unity"
            return a

        def div_linear(self, a, z):
            (gq, gr), (cq, cr) = gpu.div_linear(a[0], z), cpu.div_linear(a[1], z)
            assert np.array_equal(gr, cr), "div_linear remainder"
            return self._chk("div_linear", gq, cq), cr

        def jac_add_mixed(self, a_jac, b_aff, b_inf):
            g, c = gpu.jac_add_mixed(a_jac, b_aff, b_inf), cpu.jac_add_mixed(a_jac, b_aff, b_inf)
            ga, ca = gpu.jac_to_affine(g[None]), cpu.jac_to_affine(c[None])
            assert np.array_equal(ga[1], ca[1]) and (ca[1][0] or np.array_equal(ga[0], ca[0])), "jac_add_mixed"
            return c

        def jac_to_affine(self, jac):
            g, c = gpu.jac_to_affine(jac), cpu.jac_to_affine(jac)
            assert np.array_equal(g[1], c[1]) and np.array_equal(g[0][g[1] == 0], c[0][c[1] == 0]), "jac_to_affine"
            return c

        def commit(self, a, key="g"):
            g, c = gpu.commit(a[0], key), cpu.commit(a[1], key)
            gpu.transcript_point()          # the product backend commits asynchronously
            g = g.value
            assert np.array_equal(g[1], c[1]) and np.array_equal(g[0][g[1] == 0], c[0][c[1] == 0]), "commit"
            return c
    return Lockstep()

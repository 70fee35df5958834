import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, czk_amd, orc
from util import rand_fr_canonical, ints_to_limbs, R_MOD
ctx = czk_amd.Context(0)
g, n = 2, 10
def run(tag, bases, inf, sc):
    t = time.time()
    b = ctx.register_bases(g, bases, inf)
    got = ctx.msm(b, sc, lanes=1)
    a, i = ctx.jac_to_affine(g, got[0])
    w = orc.jac_to_affine(g, orc.msm(g, bases, inf, sc))
    print(tag, "ok" if (bool(i[0]) == w[1] and (w[1] or np.array_equal(a[0], w[0]))) else "MISMATCH", "%.2fs" % (time.time() - t), flush=True)
    b.release()
bases0 = ctx.fixed_base_points(g, rand_fr_canonical(310, n))
sc0 = rand_fr_canonical(410, n)
inf0 = np.zeros(n, np.uint8)
run("plain", bases0, inf0, sc0)
inf = inf0.copy(); inf[7] = 1
run("inf", bases0, inf, sc0)
sc = sc0.copy(); sc[0] = 0; sc[1] = ints_to_limbs([1], 4)[0]
run("zero/one scalars", bases0, inf0, sc)
bases = bases0.copy(); aw = 12
neg = bases[4].copy(); neg[aw:] = np.concatenate([orc.fq_neg(bases[4][aw:aw + 6]), orc.fq_neg(bases[4][aw + 6:])]); bases[5] = neg
sc = sc0.copy(); sc[4] = sc[5] = ints_to_limbs([R_MOD - 3], 4)[0]
run("opposite", bases, inf0, sc)
bases = bases0.copy(); bases[3] = bases[2]
sc = sc0.copy(); sc[2] = sc[3] = ints_to_limbs([5], 4)[0]
run("equal", bases, inf0, sc)

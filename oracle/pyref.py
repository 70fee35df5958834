"""Independent big-integer oracle for the BLS12-377 hot path (TEST INFRASTRUCTURE ONLY).

This file is a checker, never a product path: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  It states the *mathematics* of the path with
Python integers (no limbs, no Montgomery tricks), so that the limb-exact C restatement in
oracle/czk_oracle.c and the HIP kernels can both be pinned against something that shares no
code with either.

Reference anchors (all relative to /root/reference):
  * Fr / Fq / Fq2 constants ........ curves/bls12_377/src/fields/{fr,fq,fq2}.rs
  * G1 / G2 constants .............. curves/bls12_377/src/curves/{g1,g2}.rs
  * root-of-unity selection ........ algebra/ff/src/fields/mod.rs:337-386 (LARGE_SUBGROUP branch)
  * FFT semantics .................. algebra/poly/src/domain/radix2/{mod.rs:99-117, fft.rs:22-35}
  * coset shift = GENERATOR ........ algebra/poly/src/domain/mod.rs:139-158
  * MSM semantics (sum s_i P_i) .... algebra/ec/src/msm/variable_base.rs:12-106
  * G1 generator KAT ............... curves/bls12_377/src/curves/tests.rs:93-120

Parity status: the reference holds no golden NTT/MSM vectors; this
oracle is pinned on the reference's constant KATs (tests/test_oracle_kat.py) and is otherwise
"parity unpinned" beyond them.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------- fields
R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041  # fr.rs:33
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177  # fq.rs:26

FR_LIMBS, FQ_LIMBS = 4, 6
FR_MONT_R = (1 << 256) % R_MOD
FQ_MONT_R = (1 << 384) % Q_MOD
FR_TWO_ADICITY = 47  # fr.rs:11
FR_GENERATOR = 22  # fr.rs:69-74 decodes to 22 (the doc comment there says 11)


def limbs_to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


# constants copied as DATA (limb values) from the reference parameter files; they are checked
# for self-consistency by tests/test_oracle_kat.py
FR_MODULUS_LIMBS = [725501752471715841, 6461107452199829505, 6968279316240510977, 1345280370688173398]
FR_R_LIMBS = [9015221291577245683, 8239323489949974514, 1646089257421115374, 958099254763297437]
FR_R2_LIMBS = [2726216793283724667, 14712177743343147295, 12091039717619697043, 81024008013859129]
FR_INV = 725501752471715839
FR_GENERATOR_LIMBS = [2984901390528151251, 10561528701063790279, 5476750214495080041, 898978044469942640]
FR_TWO_ADIC_ROOT_LIMBS = [12646347781564978760, 6783048705277173164, 268534165941069093, 1121515446318641358]
FR_LARGE_SUBGROUP_ROOT_LIMBS = [0x9BFE9D90C790C167, 0x7175A69E39013BFF, 0x3FBBB698ADABCF93, 0xC59F8D8D6F0DC97]
FR_T = limbs_to_int([0xEDFDA00000021423, 0x9A3CB86F6002B354, 0xCABD34594AACC168, 0x2556])

FQ_MODULUS_LIMBS = [0x8508C00000000001, 0x170B5D4430000000, 0x1EF3622FBA094800, 0x1A22D9F300F5138F,
                    0xC63B05C06CA1493B, 0x1AE3A4617C510EA]
FQ_R_LIMBS = [202099033278250856, 5854854902718660529, 11492539364873682930, 8885205928937022213,
              5545221690922665192, 39800542322357402]
FQ_R2_LIMBS = [0xB786686C9400CD22, 0x329FCAAB00431B1, 0x22A5F11162D6B46D, 0xBFDF7D03827DC3AC,
               0x837E92F041790BF9, 0x6DFCCB1E914B88]
FQ_INV = 9586122913090633727
FQ_GENERATOR_LIMBS = [0xFC0B8000000002FA, 0x97D39CF6E000018B, 0x2072420FBFA05044, 0xCBBCBD50D97C3802,
                      0xBAF1EC35813F9EB, 0x9974A2C0945AD2]
FQ_TWO_ADIC_ROOT_LIMBS = [2022196864061697551, 17419102863309525423, 8564289679875062096,
                          17152078065055548215, 17966377291017729567, 68610905582439508]
FQ_TWO_ADICITY = 46
FQ_T = limbs_to_int([0x7510C00000021423, 0x88BEE82520005C2D, 0x67CC03D44E3C7BCD, 0x1701B28524EC688B,
                     0xE9185F1443AB18EC, 0x6B8])

FQ2_NONRESIDUE = Q_MOD - 5  # fq2.rs:13

G1_B = 1
G1_GEN = (
    81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
    241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030,
)
G1_COFACTOR = 0x170B5D4430000000 << 64  # g1.rs:27
G2_B = (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)
G2_GEN = (
    (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
     140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
    (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
     149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491),
)


def fr_to_mont(x):
    return (x * FR_MONT_R) % R_MOD


def fr_from_mont(x):
    return (x * pow(FR_MONT_R, -1, R_MOD)) % R_MOD


def fq_to_mont(x):
    return (x * FQ_MONT_R) % Q_MOD


def fq_from_mont(x):
    return (x * pow(FQ_MONT_R, -1, Q_MOD)) % Q_MOD


# ----------------------------------------------------------------------------- Fq2 (tuples)
def fq2_add(a, b):
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def fq2_mul(a, b):
    return ((a[0] * b[0] + FQ2_NONRESIDUE * a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)


def fq2_inv(a):
    n = (a[0] * a[0] - FQ2_NONRESIDUE * a[1] * a[1]) % Q_MOD
    ni = pow(n, -1, Q_MOD)
    return ((a[0] * ni) % Q_MOD, (-a[1] * ni) % Q_MOD)


# ----------------------------------------------------------------------------- curves (affine)
class _Fld:
    """Tiny field vtable so one affine-curve implementation serves G1 (Fq) and G2 (Fq2)."""

    def __init__(self, add, sub, mul, inv, zero, one):
        self.add, self.sub, self.mul, self.inv, self.zero, self.one = add, sub, mul, inv, zero, one


F1 = _Fld(lambda a, b: (a + b) % Q_MOD, lambda a, b: (a - b) % Q_MOD, lambda a, b: (a * b) % Q_MOD,
          lambda a: pow(a, -1, Q_MOD), 0, 1)
F2 = _Fld(fq2_add, fq2_sub, fq2_mul, fq2_inv, (0, 0), (1, 0))

INF = None  # point at infinity


def ec_add(F, P, Q):
    """Affine addition on y^2 = x^3 + b (a = 0), complete via explicit cases."""
    if P is INF:
        return Q
    if Q is INF:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if y1 == y2 and y1 != F.zero:
            three = F.add(F.one, F.add(F.one, F.one))
            lam = F.mul(F.mul(three, F.mul(x1, x1)), F.inv(F.add(y1, y1)))
        else:
            return INF
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F, P):
    if P is INF:
        return INF
    return (P[0], F.sub(F.zero, P[1]))


def ec_mul(F, k, P):
    acc = INF
    add = P
    while k:
        if k & 1:
            acc = ec_add(F, acc, add)
        add = ec_add(F, add, add)
        k >>= 1
    return acc


def ec_on_curve(F, P, b):
    if P is INF:
        return True
    x, y = P
    return F.mul(y, y) == F.add(F.mul(x, F.mul(x, x)), b)


def msm_naive(F, bases, scalars):
    """sum_i s_i * P_i over min(len) pairs (variable_base.rs:16)."""
    acc = INF
    for P, s in zip(bases, scalars):
        acc = ec_add(F, acc, ec_mul(F, s % R_MOD, P))
    return acc


# ----------------------------------------------------------------------------- domain / NTT
def fr_root_of_unity(log_d):
    """get_root_of_unity(2^log_d) per fields/mod.rs:337-386: LARGE^3 squared (47 - log_d) times."""
    if log_d > FR_TWO_ADICITY:
        return None
    large = fr_from_mont(limbs_to_int(FR_LARGE_SUBGROUP_ROOT_LIMBS))
    omega = pow(large, 3, R_MOD)
    for _ in range(FR_TWO_ADICITY - log_d):
        omega = omega * omega % R_MOD
    return omega


def dft(xs, log_d, inverse=False, coset=False):
    """O(D^2) definition of {fft, ifft, coset_fft, coset_ifft}_in_place on canonical integers.

    fft:        out[i] = sum_j in[j] w^{ij}                      (radix2/mod.rs:99-103)
    coset_fft:  in[j] *= g^j first, g = 22                        (domain/mod.rs:139-142)
    ifft:       out[i] = D^-1 sum_j in[j] w^{-ij}                (radix2/mod.rs:106-110)
    coset_ifft: ifft then out[i] *= g^{-i}                        (radix2/fft.rs:31-35)
    Input shorter than D is zero-extended (resize with T::zero()).
    """
    d = 1 << log_d
    assert len(xs) <= d
    xs = list(xs) + [0] * (d - len(xs))
    w = fr_root_of_unity(log_d)
    g = FR_GENERATOR
    if not inverse:
        if coset:
            xs = [x * pow(g, j, R_MOD) % R_MOD for j, x in enumerate(xs)]
        pw = [pow(w, i, R_MOD) for i in range(d)]
        return [sum(xs[j] * pw[(i * j) % d] for j in range(d)) % R_MOD for i in range(d)]
    wi = pow(w, -1, R_MOD)
    pw = [pow(wi, i, R_MOD) for i in range(d)]
    dinv = pow(d, -1, R_MOD)
    out = [sum(xs[j] * pw[(i * j) % d] for j in range(d)) * dinv % R_MOD for i in range(d)]
    if coset:
        gi = pow(g, -1, R_MOD)
        out = [x * pow(gi, i, R_MOD) % R_MOD for i, x in enumerate(out)]
    return out


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


# ----------------------------------------------------------------------------- misc helpers
def splitmix64(state):
    """One SplitMix64 step; returns (new_state, output).  Input generator per SURVEY.md section 8d."""
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def rand_fr_canonical(seed, n):
    """n canonical Fr values: 4 limbs from SplitMix64, top 3 bits masked, rejection if >= r."""
    out = []
    st = seed & 0xFFFFFFFFFFFFFFFF
    while len(out) < n:
        limbs = []
        for _ in range(4):
            st, z = splitmix64(st)
            limbs.append(z)
        limbs[3] &= 0xFFFFFFFFFFFFFFFF >> 3
        v = limbs_to_int(limbs)
        if v < R_MOD:
            out.append(v)
    return out

#!/usr/bin/env python3
"""Twisted-Edwards form of BLS12-377 G1 used INSIDE the MSM kernels (csrc/te.h), derived here from the curve equation alone.

G1 is E: y^2 = x^3 + 1 over Fq (curves/bls12_377/src/curves/g1.rs:18-23).  E has the rational 2-torsion point (-1, 0), so it is
birationally equivalent to a Montgomery curve and to a twisted Edwards curve:

    u = x + 1                      y^2 = u^3 - 3 u^2 + 3 u
    w = u / s, s = sqrt(3)         y^2 = s^3 (w^3 + A w^2 + w),  A = -3 / s        (Montgomery B v^2 = ..., v = y, B = 1 / s^3)
    xe = w / v, ye = (w - 1) / (w + 1)      a xe^2 + ye^2 = 1 + d xe^2 ye^2,  a = (A + 2) / B, d = (A - 2) / B
    X = f xe, f^2 = -a             -X^2 + ye^2 = 1 + D X^2 ye^2,  D = -d / a        (a = -1 form: 7-multiplication mixed addition)

E has full rational 2-torsion (x^2 - x + 1 splits: -3 is a square in Fq), so a d = (A^2 - 4) / B^2 = -1 / B^2 is a square and D is a
SQUARE: the unified addition law is not complete on all of E, but its exceptional cases need a point of even order (the denominators
1 +- D x1 x2 y1 y2 vanish only if P +- Q differs from the neutral element by a point of order 2 or 4), so it is exception-free on the
prime-order subgroup G1 -- where every proving-key and SRS element lives (the reference deserialises them with a subgroup check).
Handles registered for arbitrary curve points keep the XYZZ kernels (czk.h CZK_MEM_ANY_POINTS).  The script picks the
square root of 3 for which -a is a square, checks all of this with integer arithmetic against the affine group law of E, and writes
csrc/te_constants.inc (values in the kernels' residue systems).  Nothing here is reference code: the reference computes G1 MSMs in
Jacobian coordinates; this is an internal representation and every result is mapped back (parity tests compare in affine)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
GEN = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
       241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)


def inv(a):
    return pow(a % Q, Q - 2, Q)


def is_sq(a):
    return a % Q == 0 or pow(a % Q, (Q - 1) // 2, Q) == 1


def sqrt(a):
    """Tonelli-Shanks (Q - 1 = 2^46 * t)"""
    a %= Q
    assert is_sq(a)
    s, t = 0, Q - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while is_sq(z):
        z += 1
    m, c, tt, r = s, pow(z, t, Q), pow(a, t, Q), pow(a, (t + 1) // 2, Q)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x, i = x * x % Q, i + 1
        b = pow(c, 1 << (m - i - 1), Q)
        m, c, tt, r = i, b * b % Q, tt * b * b % Q, r * b % Q
    assert r * r % Q == a
    return r


def derive():
    for s in (sqrt(3), Q - sqrt(3)):
        A = -3 * inv(s) % Q
        Binv = s * s * s % Q                       # 1 / B
        a, d = (A + 2) * Binv % Q, (A - 2) * Binv % Q
        if is_sq(-a):
            f = sqrt(-a)
            D = -d * inv(a) % Q
            return {"s": s, "A": A, "a": a, "d": d, "f": f, "D": D}
    raise SystemExit("no square root of 3 makes -a a square")


C = derive()
assert is_sq(C["D"])   # see the header: complete on the odd-order subgroup only


def sw_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0:
            return None
        lam = 3 * p[0] * p[0] * inv(2 * p[1]) % Q
    else:
        lam = (q[1] - p[1]) * inv(q[0] - p[0]) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return (x, (lam * (p[0] - x) - p[1]) % Q)


def sw_to_te(p):
    """affine E -> affine (X, Y) on -X^2 + Y^2 = 1 + D X^2 Y^2; infinity -> (0, 1)"""
    if p is None:
        return (0, 1)
    w = (p[0] + 1) * inv(C["s"]) % Q
    return (C["f"] * w * inv(p[1]) % Q, (w - 1) * inv(w + 1) % Q)


def te_to_sw(t):
    X, Y = t
    if X == 0 and Y == 1:
        return None
    w = (1 + Y) * inv(1 - Y) % Q
    v = C["f"] * w * inv(X) % Q
    return ((C["s"] * w - 1) % Q, v)


def te_add(p, q):
    """the unified affine law for a = -1"""
    x1, y1 = p
    x2, y2 = q
    k = C["D"] * x1 * x2 * y1 * y2 % Q
    return ((x1 * y2 + y1 * x2) * inv(1 + k) % Q, (y1 * y2 + x1 * x2) * inv(1 - k) % Q)


def selfcheck():
    rng = random.Random(7)
    pts = [None, GEN]
    p = GEN
    for _ in range(40):
        p = sw_add(p, GEN if rng.random() < 0.5 else p)
        pts.append(p)
    for p in pts:
        t = sw_to_te(p)
        assert (-t[0] * t[0] + t[1] * t[1] - 1 - C["D"] * t[0] * t[0] * t[1] * t[1]) % Q == 0
        assert te_to_sw(t) == p
    for _ in range(200):
        p, q = rng.choice(pts), rng.choice(pts)
        assert te_add(sw_to_te(p), sw_to_te(q)) == sw_to_te(sw_add(p, q)), "TE law disagrees with the group law of E"
    for p in pts:                                  # doubling and inverse pairs through the same formula (completeness)
        assert te_add(sw_to_te(p), sw_to_te(p)) == sw_to_te(sw_add(p, p))
        neg = None if p is None else (p[0], (-p[1]) % Q)
        assert te_add(sw_to_te(p), sw_to_te(neg)) == (0, 1)
        assert sw_to_te(neg) == ((-sw_to_te(p)[0]) % Q, sw_to_te(p)[1])


def limbs(v, bits, n):
    return [(v >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def emit():
    Ru = pow(2, 392, Q)          # R' of fqu.h
    Rs = pow(2, 384, Q)          # R of field.h
    out = ["// te_constants.inc -- GENERATED by tools/gen_te_constants.py (derivation and self-check there); do not edit.",
           "// Twisted Edwards form -X^2 + Y^2 = 1 + D X^2 Y^2 of BLS12-377 G1 (E: y^2 = x^3 + 1):",
           "//   w = (x + 1) / s,  X = f w / y,  Y = (w - 1) / (w + 1),   s^2 = 3, f^2 = -(A + 2) s^3, A = -3 / s, D = (A - 2) / (A + 2)",
           "// s = %d" % C["s"], "// f = %d" % C["f"], "// D = %d (a square: the unified law is exception-free on the prime-order subgroup, see the generator)" % C["D"]]

    def arr_u(name, v, comment):
        w = limbs(v * Ru % Q, 28, 14)
        out.append("// %s  (x R' mod p, 14 x 28-bit limbs)" % comment)
        out.append("#define %s {%s}" % (name, ", ".join("0x%08xu" % x for x in w)))

    def arr_s(name, v, comment):
        w = limbs(v * Rs % Q, 32, 12)
        out.append("// %s  (x R mod p, 12 x 32-bit limbs: saturated Montgomery form)" % comment)
        out.append("#define %s {%s}" % (name, ", ".join("0x%08xu" % x for x in w)))
    arr_u("TE_2D_U", 2 * C["D"] % Q, "2 D")
    arr_u("TE_INV_D_U", inv(C["D"]), "1 / D")
    arr_s("TE_S_INV_S", inv(C["s"]), "1 / s")
    arr_s("TE_S_S", C["s"], "s")
    arr_s("TE_F_S", C["f"], "f")
    arr_s("TE_2D_S", 2 * C["D"] % Q, "2 D")
    return "\n".join(out) + "\n"


OUT = os.path.join(ROOT, "collaborative-zksnark_amd", "csrc", "te_constants.inc")

if __name__ == "__main__":
    selfcheck()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == emit() else 1)
    open(OUT, "w").write(emit())
    print("ok: s, f, D derived; TE law == group law of E on %d sums; wrote csrc/te_constants.inc" % 200)
    if "--print" in sys.argv:
        print(C)

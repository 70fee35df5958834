// curve.h -- short-Weierstrass Jacobian group law for BLS12-377 G1 (over Fq) and G2 (over Fq2), a = 0.
//
// Same formulas and the same explicit edge cases as the reference
// (algebra/ec/src/models/short_weierstrass_jacobian.rs): doubling :502-535 (a == 0 branch), mixed
// addition :570-638 (madd-2007-bl), full addition :666-728 (add-2007-bl), infinity = (1, 1, 0) :444-457.
// Group elements are compared in affine form only (Jacobian triples are not canonical), so the MSM is
// free to order its additions differently from the reference's serial loop.
#pragma once
#include "field.h"

namespace czk {

template <class F>
struct Affine {
    F x, y;
};

template <class F>
struct Jac {
    F x, y, z;
    static CZK_HD Jac zero() { return Jac{F::one(), F::one(), F::zero()}; }
    CZK_HD bool is_zero() const { return z.is_zero(); }
};

template <class F>
CZK_HD Jac<F> jac_double(const Jac<F>& p) {
    if (p.is_zero()) return p;
    F a = f_sqr(p.x);
    F b = f_sqr(p.y);
    F c = f_sqr(b);
    F d = f_dbl(f_sub(f_sub(f_sqr(f_add(p.x, b)), a), c));
    F e = f_add(a, f_dbl(a));
    F f = f_sqr(e);
    Jac<F> r;
    r.z = f_dbl(f_mul(p.z, p.y));
    r.x = f_sub(f_sub(f, d), d);
    F c8 = f_dbl(f_dbl(f_dbl(c)));
    r.y = f_sub(f_mul(f_sub(d, r.x), e), c8);
    return r;
}

// The equal-points branch of the additions is essentially never taken in an MSM (it needs two equal bases in
// one bucket).  It stays inlined: an out-of-line call here (large by-value aggregates through scratch) hung the
// G2 accumulation kernel on gfx950 / ROCm 7.2, and the cold code costs nothing when it is not executed.
template <class F>
CZK_HD Jac<F> jac_double_rare(const Jac<F>& p) {
    return jac_double(p);
}

// p + q, q affine (q_inf = q is the point at infinity)
template <class F>
CZK_HD Jac<F> jac_add_mixed(const Jac<F>& p, const Affine<F>& q, bool q_inf) {
    if (q_inf) return p;
    if (p.is_zero()) return Jac<F>{q.x, q.y, F::one()};
    F z1z1 = f_sqr(p.z);
    F u2 = f_mul(q.x, z1z1);
    F s2 = f_mul(f_mul(q.y, p.z), z1z1);
    if (p.x == u2 && p.y == s2) return jac_double_rare(p);
    // order chosen to keep few field elements live (register pressure decides this kernel's speed on gfx950):
    // Z3 first (retires Z1, Z1Z1), then I, J (retire HH, H), r (retires S2), V (retires X1), X3, Y3.
    F h = f_sub(u2, p.x);
    F hh = f_sqr(h);
    Jac<F> o;
    o.z = f_sub(f_sub(f_sqr(f_add(p.z, h)), z1z1), hh);
    F i = f_dbl(f_dbl(hh));
    F j = f_mul(h, i);
    F r = f_dbl(f_sub(s2, p.y));
    F v = f_mul(p.x, i);
    o.x = f_sub(f_sub(f_sub(f_sqr(r), j), v), v);
    F yj = f_dbl(f_mul(j, p.y));
    o.y = f_sub(f_mul(f_sub(v, o.x), r), yj);
    return o;
}

template <class F>
CZK_HD Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
    if (p.is_zero()) return q;
    if (q.is_zero()) return p;
    // same register-pressure-aware ordering as jac_add_mixed: every product retires an input as early as possible
    F z1z1 = f_sqr(p.z);
    F z2z2 = f_sqr(q.z);
    F u1 = f_mul(p.x, z2z2);
    F u2 = f_mul(q.x, z1z1);
    F s1 = f_mul(f_mul(p.y, q.z), z2z2);
    F s2 = f_mul(f_mul(q.y, p.z), z1z1);
    if (u1 == u2 && s1 == s2) return jac_double_rare(p);
    F h = f_sub(u2, u1);
    Jac<F> o;
    o.z = f_mul(f_sub(f_sub(f_sqr(f_add(p.z, q.z)), z1z1), z2z2), h);
    F i = f_sqr(f_dbl(h));
    F j = f_mul(h, i);
    F r = f_dbl(f_sub(s2, s1));
    F v = f_mul(u1, i);
    o.x = f_sub(f_sub(f_sqr(r), j), f_dbl(v));
    o.y = f_sub(f_mul(r, f_sub(v, o.x)), f_dbl(f_mul(s1, j)));
    return o;
}

// From<Projective> for Affine (short_weierstrass_jacobian.rs:768-789); returns the infinity flag.
template <class F>
CZK_HD bool jac_to_affine(const Jac<F>& p, Affine<F>& out) {
    if (p.is_zero()) {
        out.x = F::zero();
        out.y = F::one();
        return true;
    }
    F zi = f_inv(p.z);
    F zi2 = f_sqr(zi);
    out.x = f_mul(p.x, zi2);
    out.y = f_mul(p.y, f_mul(zi2, zi));
    return false;
}

template <class F>
CZK_HD Affine<F> aff_load(const u64* p) {
    return Affine<F>{FieldIO<F>::load(p), FieldIO<F>::load(p + FieldIO<F>::W64)};
}
template <class F>
CZK_HD void aff_store(u64* p, const Affine<F>& a) {
    FieldIO<F>::store(p, a.x);
    FieldIO<F>::store(p + FieldIO<F>::W64, a.y);
}
template <class F>
CZK_HD Jac<F> jac_load(const u64* p) {
    constexpr int W = FieldIO<F>::W64;
    return Jac<F>{FieldIO<F>::load(p), FieldIO<F>::load(p + W), FieldIO<F>::load(p + 2 * W)};
}
template <class F>
CZK_HD void jac_store(u64* p, const Jac<F>& a) {
    constexpr int W = FieldIO<F>::W64;
    FieldIO<F>::store(p, a.x);
    FieldIO<F>::store(p + W, a.y);
    FieldIO<F>::store(p + 2 * W, a.z);
}

// ---------------------------------------------------------------------------------------------
// XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ == 0) for the MSM's internal bucket
// arithmetic: mixed addition 8M + 2S and general addition 12M + 2S (EFD madd-2008-s / add-2008-s, a = 0), against
// 7M + 4S / 11M + 5S for the Jacobian forms above.  Only internal: every result leaves the library as the
// reference's Jacobian triple (xyzz_to_jac) and is compared in affine form.  Edge cases mirror
// short_weierstrass_jacobian.rs:570-597 / :666-689: infinity operands, equal points (-> doubling), opposite
// points (-> infinity).
// ---------------------------------------------------------------------------------------------
template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    static CZK_HD XYZZ zero() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
    CZK_HD bool is_zero() const { return zz.is_zero(); }
};

// 2 * (x, y) for an affine point (mdbl-2008-s-1, a = 0)
template <class F>
CZK_HD XYZZ<F> xyzz_double_affine(const Affine<F>& q) {
    F u = f_dbl(q.y);
    F v = f_sqr(u);
    F w = f_mul(u, v);
    F s = f_mul(q.x, v);
    F xx = f_sqr(q.x);
    F m = f_add(f_dbl(xx), xx);
    XYZZ<F> o;
    o.x = f_sub(f_sqr(m), f_dbl(s));
    o.y = f_sub(f_mul(m, f_sub(s, o.x)), f_mul(w, q.y));
    o.zz = v;
    o.zzz = w;
    return o;
}
// 2 * p (dbl-2008-s-1, a = 0)
template <class F>
CZK_HD XYZZ<F> xyzz_double(const XYZZ<F>& p) {
    if (p.is_zero()) return p;
    F u = f_dbl(p.y);
    F v = f_sqr(u);
    F w = f_mul(u, v);
    F s = f_mul(p.x, v);
    F xx = f_sqr(p.x);
    F m = f_add(f_dbl(xx), xx);
    XYZZ<F> o;
    o.x = f_sub(f_sqr(m), f_dbl(s));
    o.y = f_sub(f_mul(m, f_sub(s, o.x)), f_mul(w, p.y));
    o.zz = f_mul(v, p.zz);
    o.zzz = f_mul(w, p.zzz);
    return o;
}

// p + q, q affine and not infinity (madd-2008-s); product order keeps few values live
template <class F>
CZK_HD XYZZ<F> xyzz_add_mixed(const XYZZ<F>& p, const Affine<F>& q) {
    if (p.is_zero()) return XYZZ<F>{q.x, q.y, F::one(), F::one()};
    F pp = f_sub(f_mul(q.x, p.zz), p.x);        // P = U2 - X1
    F r = f_sub(f_mul(q.y, p.zzz), p.y);        // R = S2 - Y1
    if (pp.is_zero()) {
        if (r.is_zero()) return xyzz_double_affine(q);   // same point
        return XYZZ<F>::zero();                          // opposite points
    }
    F p2 = f_sqr(pp);                            // PP
    XYZZ<F> o;
    o.zz = f_mul(p.zz, p2);
    F p3 = f_mul(pp, p2);                        // PPP
    o.zzz = f_mul(p.zzz, p3);
    F qv = f_mul(p.x, p2);                       // Q
    o.x = f_sub(f_sub(f_sqr(r), p3), f_dbl(qv));
    o.y = f_sub(f_mul(r, f_sub(qv, o.x)), f_mul(p.y, p3));
    return o;
}

// In-place form of xyzz_add_mixed for hot loops: the accumulator is four separate values updated in place (a
// struct returned from several exits made hipcc keep X and Y in a stack slot across iterations: 24 scratch stores and
// loads per addition).
template <class F>
CZK_HD void xyzz_acc_mixed(F& ax, F& ay, F& azz, F& azzz, const F& qx, const F& qy) {
    if (azz.is_zero()) {
        ax = qx;
        ay = qy;
        azz = F::one();
        azzz = F::one();
        return;
    }
    F pp = f_sub(f_mul(qx, azz), ax);
    F r = f_sub(f_mul(qy, azzz), ay);
    if (pp.is_zero()) {
        XYZZ<F> d = r.is_zero() ? xyzz_double_affine(Affine<F>{qx, qy}) : XYZZ<F>::zero();
        ax = d.x;
        ay = d.y;
        azz = d.zz;
        azzz = d.zzz;
        return;
    }
    F p2 = f_sqr(pp);
    azz = f_mul(azz, p2);
    F p3 = f_mul(pp, p2);
    azzz = f_mul(azzz, p3);
    F qv = f_mul(ax, p2);
    ax = f_sub(f_sub(f_sqr(r), p3), f_dbl(qv));
    ay = f_sub(f_mul(r, f_sub(qv, ax)), f_mul(ay, p3));
}

// p + q (add-2008-s)
template <class F>
CZK_HD XYZZ<F> xyzz_add(const XYZZ<F>& p, const XYZZ<F>& q) {
    if (p.is_zero()) return q;
    if (q.is_zero()) return p;
    F u1 = f_mul(p.x, q.zz);
    F pp = f_sub(f_mul(q.x, p.zz), u1);          // P = U2 - U1
    F s1 = f_mul(p.y, q.zzz);
    F r = f_sub(f_mul(q.y, p.zzz), s1);          // R = S2 - S1
    if (pp.is_zero()) {
        if (r.is_zero()) return xyzz_double(p);
        return XYZZ<F>::zero();
    }
    F p2 = f_sqr(pp);
    F p3 = f_mul(pp, p2);
    XYZZ<F> o;
    o.zz = f_mul(f_mul(p.zz, q.zz), p2);
    o.zzz = f_mul(f_mul(p.zzz, q.zzz), p3);
    F qv = f_mul(u1, p2);
    o.x = f_sub(f_sub(f_sqr(r), p3), f_dbl(qv));
    o.y = f_sub(f_mul(r, f_sub(qv, o.x)), f_mul(s1, p3));
    return o;
}

// (X, Y, ZZ, ZZZ) -> the Jacobian triple (X ZZ, Y ZZZ, ZZ): x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 = Y / ZZZ
template <class F>
CZK_HD Jac<F> xyzz_to_jac(const XYZZ<F>& p) {
    if (p.is_zero()) return Jac<F>::zero();
    return Jac<F>{f_mul(p.x, p.zz), f_mul(p.y, p.zzz), p.zz};
}

template <class F>
CZK_HD XYZZ<F> xyzz_load(const u64* p) {
    constexpr int W = FieldIO<F>::W64;
    return XYZZ<F>{FieldIO<F>::load(p), FieldIO<F>::load(p + W), FieldIO<F>::load(p + 2 * W), FieldIO<F>::load(p + 3 * W)};
}
template <class F>
CZK_HD void xyzz_store(u64* p, const XYZZ<F>& a) {
    constexpr int W = FieldIO<F>::W64;
    FieldIO<F>::store(p, a.x);
    FieldIO<F>::store(p + W, a.y);
    FieldIO<F>::store(p + 2 * W, a.zz);
    FieldIO<F>::store(p + 3 * W, a.zzz);
}

typedef Affine<Fq> G1Affine;
typedef Jac<Fq> G1Jac;
typedef Affine<Fq2> G2Affine;
typedef Jac<Fq2> G2Jac;

}  // namespace czk

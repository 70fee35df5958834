// fq2pu.h -- G2 points for the bucket REDUCTION: Fq2 in the unsaturated residue system of fqu.h, one element spread over a PAIR of
// adjacent lanes (device only).
//
// The G2 reduction used to run on fq2p.h's saturated lane pairs: 2 Montgomery products of ~620 instructions per lane and Fq2 product,
// 9 % of a proof's VALU instructions (profiles/r03_valu_share.txt) for 2.3 additions per bucket.  Here lane 2k holds the c0 halves and
// lane 2k + 1 the c1 halves as 14 x 28-bit limbs, and an Fq2 product is ONE fqu_mul_add per lane -- two limb products under a single
// Montgomery reduction:
//     even lane: c0 = a0 b0 + a1 (K p - 5 b1)          odd lane: c1 = a1 b0 + a0 b1            (u^2 = -5, quadratic_extension.rs:571-583)
// i.e. "own * X + other * Y" with (X, Y) prepared once per multiplier (p2_b) and reused by every product that shares it.  Squarings go
// through the same product (the complex-squaring form would need a second exchange).  Halves travel with DPP quad_perm [1,0,3,2].
// Values are the same field elements as Fq2's; coordinates are kept in "u-form" (value * 2^392 mod p + k p, packed 12 x u32 per half)
// from k_accumulate_u2's store to the last step, where the result is converted to the reference's Jacobian triple in Montgomery form.
//
// Value discipline (units of p; every value normalised, limbs < 2^28): inputs x < 100, y < 36, zz, zzz < 3.2 (k_accumulate_u2's
// buckets: x < 85, y < 36, zz, zzz < 3; sums of this file: x < 9.2, the rest < 1.2).  A product's output is below
// 1.01 + (|a| |b| + |a| |Y|) / 38968 with |Y| <= 16 (small multipliers, < 3.2) or 512 (multipliers up to 102): < 1.2 everywhere but
// for V = (2 Y1)^2 (< 2.1) and X1^2 (< 2.6) of the doubling.
#pragma once
#include "fq2p.h"
#include "fqu.h"

namespace czk {

__device__ __forceinline__ FqU pair_swap_u(const FqU& a) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = pair_swap_u32(a.l[i]);
    return r;
}
__device__ __forceinline__ FqU fqu_select(bool c, const FqU& a, const FqU& b) {   // c ? a : b
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
struct P2A {   // multiplicand: this lane's half and the partner's; limbs < 2^30
    FqU own, oth;
};
struct P2B {   // multiplier, normalised: even lane (b0, K p - 5 b1), odd lane (b0, b1) swapped into product order
    FqU x, y;
};
__device__ __forceinline__ P2A p2_a(const FqU& a) { return P2A{a, pair_swap_u(a)}; }
// BIG: the multiplier's value may reach 102 p (K = 512); otherwise it is below 3.2 p (K = 16)
template <bool BIG>
__device__ __forceinline__ P2B p2_b(const FqU& b) {
    const bool par = pair_parity();
    const FqU oth = pair_swap_u(b);
    const FqU n5 = fqu_neg5<BIG>(oth);
    return P2B{fqu_select(par, oth, b), fqu_select(par, b, n5)};
}
__device__ __forceinline__ FqU p2_mul(const P2A& a, const P2B& b) { return fqu_mul_add(a.own, b.x, a.oth, b.y); }
__device__ __forceinline__ bool pair_all(bool c) {
    const u32 v = c ? 1u : 0u;
    return (v & pair_swap_u32(v)) != 0;
}

struct XYZZU2 {
    FqU x, y, zz, zzz;   // this lane's halves
    bool inf;            // the same on both lanes of a pair; stored as zz == 0 exactly
};
__device__ __forceinline__ XYZZU2 xyzzu2_zero() {
    XYZZU2 r;
    r.inf = true;
#pragma unroll
    for (int i = 0; i < 14; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    return r;
}
// memory: an XYZZ point over Fq2 is (x.c0, x.c1, y.c0, ...), 6 u64 per half: this lane touches words [12 k + 6 parity, + 6)
__device__ __forceinline__ XYZZU2 xyzzu2_load(const u64* p) {
    p += pair_parity() ? 6 : 0;
    const Fq x = fp_load<FqParams>(p), y = fp_load<FqParams>(p + 12), zz = fp_load<FqParams>(p + 24), zzz = fp_load<FqParams>(p + 36);
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= zz.l[i];
    XYZZU2 r;
    r.inf = pair_all(any == 0);
    r.x = fqu_unpack(x);
    r.y = fqu_unpack(y);
    r.zz = fqu_unpack(zz);
    r.zzz = fqu_unpack(zzz);
    return r;
}
__device__ __forceinline__ void xyzzu2_store(u64* p, const XYZZU2& a) {
    p += pair_parity() ? 6 : 0;
    if (a.inf) {
        const Fq z = Fq::zero();
        fp_store<FqParams>(p, z);
        fp_store<FqParams>(p + 12, z);
        fp_store<FqParams>(p + 24, z);
        fp_store<FqParams>(p + 36, z);
        return;
    }
    fp_store<FqParams>(p, fqu_pack(a.x));
    fp_store<FqParams>(p + 12, fqu_pack(a.y));
    fp_store<FqParams>(p + 24, fqu_pack(a.zz));
    fp_store<FqParams>(p + 36, fqu_pack(a.zzz));
}
// u-form <-> fq2p.h's saturated lane pairs (rare paths and the final result); conversions are linear, so they act on the halves
__device__ __forceinline__ XYZZ<Fq2P> xyzzu2_to_sat(const XYZZU2& a) {
    if (a.inf) return XYZZ<Fq2P>::zero();
    const Fq kf = fqu_k_from_u();
    return XYZZ<Fq2P>{Fq2P{fp_mul(fqu_pack(a.x), kf)}, Fq2P{fp_mul(fqu_pack(a.y), kf)}, Fq2P{fp_mul(fqu_pack(a.zz), kf)}, Fq2P{fp_mul(fqu_pack(a.zzz), kf)}};
}
__device__ __forceinline__ XYZZU2 xyzzu2_from_sat(const XYZZ<Fq2P>& s) {
    if (s.is_zero()) return xyzzu2_zero();
    const Fq kt = fqu_k_to_u();
    XYZZU2 r;
    r.inf = false;
    r.x = fqu_unpack(fp_mul(s.x.h, kt));
    r.y = fqu_unpack(fp_mul(s.y.h, kt));
    r.zz = fqu_unpack(fp_mul(s.zz.h, kt));
    r.zzz = fqu_unpack(fp_mul(s.zzz.h, kt));
    return r;
}
// the complete formulas of curve.h on saturated lane pairs; both lanes of a pair always arrive here together
__device__ __noinline__ XYZZU2 xyzzu2_add_slow(XYZZU2 a, XYZZU2 b) { return xyzzu2_from_sat(xyzz_add(xyzzu2_to_sat(a), xyzzu2_to_sat(b))); }
__device__ __noinline__ XYZZU2 xyzzu2_double_slow(XYZZU2 a) { return xyzzu2_from_sat(xyzz_double(xyzzu2_to_sat(a))); }

// a += b (add-2008-s, 12M + 2S over Fq2 = 14 products per lane + the fused Y3).  H = U2 - U1 + 4 p: each half lies in (2.8 p, 5.2 p) and
// is 0 mod p only if it equals j p, j in 3..5 -- the low limb is then j (p == 1 mod 2^28).  H == 0 in Fq2 needs both halves: the
// suspicious pairs (and every genuine P == +-Q) take the saturated complete formulas.
__device__ __forceinline__ void xyzzu2_add(XYZZU2& a, const XYZZU2& b) {
    if (b.inf) return;
    if (a.inf) {
        a = b;
        return;
    }
    // (products are ordered so that each prepared multiplier dies early: the kernels around this hold two more points in registers)
    const P2B bzz = p2_b<false>(b.zz);
    const FqU u1 = p2_mul(p2_a(a.x), bzz);
    const FqU u2 = p2_mul(p2_a(b.x), p2_b<false>(a.zz));
    const FqU pp = fqu_normalize(fqu_sub_lazy<4>(u2, u1));
    if (pair_all((pp.l[0] - 3u) <= 2u)) {
        a = xyzzu2_add_slow(a, b);
        return;
    }
    const FqU zzab = p2_mul(p2_a(a.zz), bzz);
    const P2B bzzz = p2_b<false>(b.zzz);
    const FqU s1 = p2_mul(p2_a(a.y), bzzz);
    const FqU zzzab = p2_mul(p2_a(a.zzz), bzzz);
    const FqU s2 = p2_mul(p2_a(b.y), p2_b<false>(a.zzz));
    const FqU r = fqu_normalize(fqu_sub_lazy<4>(s2, s1));
    const P2A ppa = p2_a(pp);
    const FqU p2 = p2_mul(ppa, p2_b<true>(pp));
    const P2B p2b = p2_b<false>(p2);
    const FqU p3 = p2_mul(ppa, p2b);
    const FqU qv = p2_mul(p2_a(u1), p2b);
    a.zz = p2_mul(p2_a(zzab), p2b);
    const P2B p3b = p2_b<false>(p3);
    a.zzz = p2_mul(p2_a(zzzab), p3b);
    const P2A ra = p2_a(r);
    const FqU t = p2_mul(ra, p2_b<true>(r));
    a.x = fqu_sub3_norm(t, p3, qv);                                      // R^2 - PPP - 2 Q + 8 p  < 9.2 p
    const P2B db = p2_b<true>(fqu_normalize(fqu_sub_lazy<16>(qv, a.x)));  // Q - X3 + 16 p  < 17.2 p
    FqU ns1;                                                             // 8 p - S1 (lazy)
#pragma unroll
    for (int i = 0; i < 14; i++) ns1.l[i] = fqu_8p(i) - s1.l[i];
    const P2A nsa = p2_a(ns1);
    a.y = fqu_mul_add4(ra.own, db.x, ra.oth, db.y, nsa.own, p3b.x, nsa.oth, p3b.y);   // R (Q - X3) - S1 PPP, one reduction
}
// a = 2 a (dbl-2008-s-1, a = 0).  U = 2 Y1 == 0 (a point of order two) goes through the complete formulas, which return infinity.
__device__ __forceinline__ void xyzzu2_double(XYZZU2& a) {
    if (a.inf) return;
    FqU u;
#pragma unroll
    for (int i = 0; i < 14; i++) u.l[i] = a.y.l[i] + a.y.l[i];
    u = fqu_normalize(u);                                                // < 72 p
    if (pair_all(u.l[0] < 72u)) {
        a = xyzzu2_double_slow(a);
        return;
    }
    const P2A ua = p2_a(u);
    const FqU v = p2_mul(ua, p2_b<true>(u));                             // < 2.1 p
    const P2B vb = p2_b<false>(v);
    const FqU w = p2_mul(ua, vb);
    const P2A xa = p2_a(a.x);
    const FqU s = p2_mul(xa, vb);
    const FqU xx = p2_mul(xa, p2_b<true>(a.x));                          // < 2.6 p
    FqU m;
#pragma unroll
    for (int i = 0; i < 14; i++) m.l[i] = 3u * xx.l[i];
    m = fqu_normalize(m);                                                // M = 3 X1^2 < 7.8 p
    const P2A ma = p2_a(m);
    const FqU mm = p2_mul(ma, p2_b<true>(m));
    FqU x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = mm.l[i] + (fqu_8p_wide(i) - s.l[i] - s.l[i]);   // M^2 - 2 S + 8 p
    x3 = fqu_normalize(x3);
    const P2B db = p2_b<true>(fqu_normalize(fqu_sub_lazy<16>(s, x3)));  // S - X3 + 16 p
    FqU nw;
#pragma unroll
    for (int i = 0; i < 14; i++) nw.l[i] = fqu_8p(i) - w.l[i];           // 8 p - W (lazy)
    const P2A nwa = p2_a(nw);
    const P2B yb = p2_b<true>(a.y);
    const FqU y3 = fqu_mul_add4(ma.own, db.x, ma.oth, db.y, nwa.own, yb.x, nwa.oth, yb.y);   // M (S - X3) - W Y1
    a.zz = p2_mul(p2_a(v), p2_b<false>(a.zz));
    a.zzz = p2_mul(p2_a(w), p2_b<false>(a.zzz));
    a.x = x3;
    a.y = y3;
}


// acc += (+-)P, P an affine table point (madd-2008-s over Fq2, 8 products per lane + the fused Y3): the bucket ACCUMULATION on lane pairs
// (k_accumulate_u2p, CZK_G2_MODE=1 / 2).  One lane's footprint is 204 registers against the single-lane kernel's 304 (fqu.h fq2u_xyzz_acc_mixed, one
// wave per SIMD), so two waves share a SIMD.  MEASURED: alone 23.9 against 29.3 ms per 2^20-point 4-lane launch, 25.3 - 25.9 against 33.1 - 33.9 ms
// inside the Groth16 pipeline -- and the PROOF gets slower (78.7 - 81.3 against 76.2 - 77.6 ms, any MSM order): two such waves leave no room for
// the NTT / sort / reduction waves that run beside the single-lane kernel's one wave, the accumulate stream then waits for them, and the machine's
// instruction throughput was already used either way (profiles/r03_g2_lane_pairs.txt).  Not adopted; the default stays k_accumulate_u2.
// Bounds (units of p): ax < 9.2 after an addition (a first point's < 1), ay < 4, azz, azzz < 3.2; qx canonical; qy canonical or the lazy
// 4 p - y of a negated point.  H = U2 - X1 + 16 p: each half in (6.8, 17.2); it is 0 mod p only if it equals j p, j in 7..17, i.e. its low limb
// is j -- returns false when both halves look like that (the caller defers the point to the saturated complete formulas).
__device__ __forceinline__ bool xyzzu2_acc_mixed(FqU& ax, FqU& ay, FqU& azz, FqU& azzz, const FqU& qx, const FqU& qy) {
    const FqU u2 = p2_mul(p2_a(qx), p2_b<false>(azz));
    const FqU pp = fqu_normalize(fqu_sub_lazy<16>(u2, ax));
    if (pair_all((pp.l[0] - 6u) <= 12u)) return false;
    const FqU s2 = p2_mul(p2_a(qy), p2_b<false>(azzz));
    const FqU r = fqu_normalize(fqu_sub_lazy<8>(s2, ay));                 // (4, 9.2)
    const P2A ppa = p2_a(pp);
    const FqU p2 = p2_mul(ppa, p2_b<true>(pp));
    const P2B p2b = p2_b<false>(p2);
    const FqU zz3 = p2_mul(p2_a(azz), p2b);
    const FqU p3 = p2_mul(ppa, p2b);
    const FqU qv = p2_mul(p2_a(ax), p2b);
    const P2B p3b = p2_b<false>(p3);
    const FqU zzz3 = p2_mul(p2_a(azzz), p3b);
    const P2A ra = p2_a(r);
    const FqU t = p2_mul(ra, p2_b<true>(r));
    const FqU x3 = fqu_sub3_norm(t, p3, qv);                              // R^2 - PPP - 2 Q + 8 p  < 9.2 p
    const P2B db = p2_b<true>(fqu_normalize(fqu_sub_lazy<16>(qv, x3)));   // Q - X3 + 16 p  < 17.2 p
    FqU nay;                                                              // 8 p - Y1 (lazy)
#pragma unroll
    for (int i = 0; i < 14; i++) nay.l[i] = fqu_8p(i) - ay.l[i];
    const P2A na = p2_a(nay);
    ay = fqu_mul_add4(ra.own, db.x, ra.oth, db.y, na.own, p3b.x, na.oth, p3b.y);   // R (Q - X3) - Y1 PPP, one reduction
    ax = x3;
    azz = zz3;
    azzz = zzz3;
    return true;
}

}  // namespace czk

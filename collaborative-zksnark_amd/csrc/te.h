// te.h -- BLS12-377 G1 in twisted Edwards form for the MSM's bucket kernels (device only, G1 only).
//
// G1 is E: y^2 = x^3 + 1 (curves/bls12_377/src/curves/g1.rs:18-23).  E has rational 2-torsion, so it is birationally equivalent to
//     -X^2 + Y^2 = 1 + D X^2 Y^2        (tools/gen_te_constants.py: derivation, constants, integer self-check against the group law of E)
// and in extended coordinates (X : Y : Z : T), X Y = Z T, a point of a precomputed table -- kept as (Y - X, Y + X, 2 D X Y) -- is
// added with SEVEN multiplications and no squarings (madd-2008-hwcd-3) against 8M + 2S for the XYZZ mixed addition: 2 646 instead of
// 3 416 v_mad_u64_u32, and every output is a multiply result, so the lazy sums in between need no normalisation.  The law is unified
// (doubling, inverse pairs and the neutral element go through the same formula) and exception-free on the prime-order subgroup:
// no exception list, no dirty flags, no fix-up kernels.  D is a square (E has full 2-torsion), so exceptional cases exist for points
// of even order: handles whose bases are arbitrary curve points keep the XYZZ kernels (czk.h CZK_MEM_ANY_POINTS), and registration
// falls back to them by itself when a base has no image under the map (y = 0 or w = -1).
//
// The representation is internal: tables are converted once at registration (k_sw_to_te_niels), buckets and every array of the
// reduction hold extended coordinates in the unsaturated residue system of fqu.h ("u-form", 4 x 48 bytes like an XYZZ bucket), and
// the last reduction step maps the result back to the reference's Jacobian triple WITHOUT an inversion (teu_to_jac).  The reference
// computes G1 MSMs in Jacobian coordinates (algebra/ec/src/msm/variable_base.rs:12-106); results are compared in affine, as ever.
#pragma once
#include "fqu.h"
#ifdef CZK_TE_IL   // A/B builds of the lab library only
#include "lab/fqu_il.h"
#endif
#include "te_constants.inc"

namespace czk {

struct TEU {
    FqU x, y, z, t;   // normalised limbs; multiply outputs (< 1.01 p) except after teu_from_niels (x < 5 p, y < 2 p, z = 2)
};
__device__ __forceinline__ FqU fqu_const(const u32 (&m)[14]) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = m[i];
    return r;
}
__device__ __forceinline__ FqU te_2d_u() {
    constexpr u32 m[14] = TE_2D_U;
    return fqu_const(m);
}
__device__ __forceinline__ FqU fqu_zero() {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ TEU teu_identity() { return TEU{fqu_zero(), fqu_one(), fqu_one(), fqu_zero()}; }
__device__ __forceinline__ TEU teu_load(const u64* p) {
    return TEU{fqu_unpack(fp_load<FqParams>(p)), fqu_unpack(fp_load<FqParams>(p + 6)), fqu_unpack(fp_load<FqParams>(p + 12)),
               fqu_unpack(fp_load<FqParams>(p + 18))};
}
__device__ __forceinline__ void teu_store(u64* p, const TEU& a) {
    fp_store<FqParams>(p, fqu_pack(a.x));
    fp_store<FqParams>(p + 6, fqu_pack(a.y));
    fp_store<FqParams>(p + 12, fqu_pack(a.z));
    fp_store<FqParams>(p + 18, fqu_pack(a.t));
}

// the four output products shared by every formula below: (E, F, G, H) -> (E F, G H, F G, E H); operands lazy (limbs < 2^30)
__device__ __forceinline__ void teu_finish(TEU& a, const FqU& E, const FqU& F, const FqU& G, const FqU& H) {
    a.x = fqu_mul(E, F);
    a.y = fqu_mul(G, H);
    a.z = fqu_mul(F, G);
    a.t = fqu_mul(E, H);
}
// a += (ym, yp, k2) = (Y2 - X2, Y2 + X2, 2 D X2 Y2) of an affine table point; madd-2008-hwcd-3, 7M.  ym, yp canonical (< p);
// k2 canonical or the lazy 4 p - k2 of a negated point (limbs < 2^30).
__device__ __forceinline__ void teu_madd(TEU& a, const FqU& ym, const FqU& yp, const FqU& k2) {
    const FqU A = fqu_mul(fqu_sub_lazy<8>(a.y, a.x), ym);   // (Y1 - X1 + 8 p)(Y2 - X2)   (X1 < 5 p after teu_from_niels)
    const FqU B = fqu_mul(fqu_add_lazy(a.y, a.x), yp);      // (Y1 + X1)(Y2 + X2)
    const FqU C = fqu_mul(a.t, k2);                         // T1 2 D T2
    FqU F, G;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const u32 d = a.z.l[i] + a.z.l[i];                  // D = 2 Z1
        F.l[i] = d + (fqu_4p(i) - C.l[i]);                  // D - C + 4 p, limbs < 2^30
        G.l[i] = d + C.l[i];
    }
    teu_finish(a, fqu_sub_lazy<4>(B, A), F, G, fqu_add_lazy(B, A));
}
#ifdef CZK_TE_IL
// The same addition with its products run side by side (fqu_il.h): A, B, C as three interleaved chains, then the four output
// products as four.  Same values, same instruction count; the lane has 3 - 4 multiply-add chains in flight instead of one.
__device__ __forceinline__ void teu_madd_il(TEU& a, const FqU& ym, const FqU& yp, const FqU& k2) {
    const FqU u = fqu_sub_lazy<8>(a.y, a.x), v = fqu_add_lazy(a.y, a.x);
    FqU abc[3];
    {
        const FqU* const x[3][1] = {{&u}, {&v}, {&a.t}};
        const FqU* const y[3][1] = {{&ym}, {&yp}, {&k2}};
        fqu_mul_il<3, 1>(x, y, abc);
    }
    FqU F, G;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const u32 d = a.z.l[i] + a.z.l[i];
        F.l[i] = d + (fqu_4p(i) - abc[2].l[i]);
        G.l[i] = d + abc[2].l[i];
    }
    const FqU E = fqu_sub_lazy<4>(abc[1], abc[0]), H = fqu_add_lazy(abc[1], abc[0]);
    FqU o[4];
    {
        const FqU* const x[4][1] = {{&E}, {&G}, {&F}, {&E}};
        const FqU* const y[4][1] = {{&F}, {&H}, {&G}, {&H}};
        fqu_mul_il<4, 1>(x, y, o);
    }
    a.x = o[0];
    a.y = o[1];
    a.z = o[2];
    a.t = o[3];
}
#endif
// a += b, both extended (add-2008-hwcd-3, 8M + one multiplication by the constant 2 D)
__device__ __forceinline__ void teu_add(TEU& a, const TEU& b) {
    const FqU A = fqu_mul(fqu_sub_lazy<8>(a.y, a.x), fqu_sub_lazy<8>(b.y, b.x));
    const FqU B = fqu_mul(fqu_add_lazy(a.y, a.x), fqu_add_lazy(b.y, b.x));
    const FqU C = fqu_mul(fqu_mul(a.t, b.t), te_2d_u());
    const FqU Dh = fqu_mul(a.z, b.z);
    FqU F, G;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const u32 d = Dh.l[i] + Dh.l[i];
        F.l[i] = d + (fqu_4p(i) - C.l[i]);
        G.l[i] = d + C.l[i];
    }
    teu_finish(a, fqu_sub_lazy<4>(B, A), F, G, fqu_add_lazy(B, A));
}
// a = 2 a (dbl-2008-hwcd with a = -1: 4M + 4S)
__device__ __forceinline__ void teu_double(TEU& a) {
    const FqU A = fqu_sqr(a.x), B = fqu_sqr(a.y), Cz = fqu_sqr(a.z);
    const FqU S = fqu_sqr(fqu_add_lazy(a.x, a.y));
    const FqU G = fqu_normalize(fqu_sub_lazy<4>(B, A));     // G = B - A + 4 p, normalised so that F below stays under 2^30 per limb
    FqU E, F, H;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        E.l[i] = S.l[i] + (fqu_8p_u2(i) - A.l[i] - B.l[i]);             // (X + Y)^2 - A - B + 8 p
        F.l[i] = G.l[i] + (fqu_8p_u2(i) - Cz.l[i] - Cz.l[i]);           // G - 2 Z^2 + 8 p
        H.l[i] = fqu_8p_u2(i) - A.l[i] - B.l[i];                        // -A - B + 8 p
    }
    teu_finish(a, E, F, G, H);
}

// (X : Y : Z : T) -> the reference's Jacobian triple of the corresponding point of E, without an inversion:
//   w = (Z + Y) / (Z - Y),  x = s w - 1,  y = f w Z / X;   with Zj = (Z - Y) X:  Xj = (s (Z + Y) - (Z - Y)) (Z - Y) X^2,
//   Yj = f (Z + Y) Z (Z - Y)^2 X^2.  The neutral element (0 : 1 : 1 : 0) gives Zj = 0, the reference's point at infinity.
__device__ __forceinline__ Jac<Fq> teu_to_jac(const TEU& a) {
    constexpr u32 sm[12] = TE_S_S, fm[12] = TE_F_S;
    Fq s, f;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        s.l[i] = sm[i];
        f.l[i] = fm[i];
    }
    const Fq kf = fqu_k_from_u();
    const Fq X = fp_mul(fqu_pack(a.x), kf), Y = fp_mul(fqu_pack(a.y), kf), Z = fp_mul(fqu_pack(a.z), kf);
    const Fq zp = fp_add(Z, Y), zm = fp_sub(Z, Y);
    const Fq x2 = fp_sqr(X), zmx2 = fp_mul(zm, x2);
    Jac<Fq> r;
    r.z = fp_mul(zm, X);
    r.x = fp_mul(fp_sub(fp_mul(s, zp), zm), zmx2);
    r.y = fp_mul(fp_mul(fp_mul(f, zp), Z), fp_mul(zm, zmx2));
    if (r.z.is_zero()) return Jac<Fq>::zero();
    return r;
}

// One table point: 48 u32 = (Y - X, Y + X, 2 D X Y), each x R' mod p as 14 limbs of 28 bits in 16 u32 (two pad words): 192 bytes,
// three aligned 64-byte sectors -- the limbs are loaded as they are used, no unpacking in the hot loop.
constexpr int TE_POINT_U64 = 24;
__device__ __forceinline__ FqU te_load_coord(const u64* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    FqU r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void te_store_coord(u64* p, const FqU& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}
// `neg` adds -P = (-X, Y): the roles of Y - X and Y + X swap, 2 D X Y changes sign.  (Grouping a bucket's entries by sign so that the
// sign could be compiled into two loops was tried: the lanes of a wave then run max(#positive) + max(#negative) iterations instead of
// max(#entries) -- 8.76 against 7.98 ms per launch.)
__device__ __forceinline__ void te_load_niels(const u64* pp, bool neg, FqU& ym, FqU& yp, FqU& k2) {
    const FqU a = te_load_coord(pp), b = te_load_coord(pp + 8);
    k2 = te_load_coord(pp + 16);
    ym = neg ? b : a;
    yp = neg ? a : b;
    if (neg) {
#pragma unroll
        for (int i = 0; i < 14; i++) k2.l[i] = fqu_4p(i) - k2.l[i];      // 4 p - k, lazy
    }
}
// a = +-P for the first entry of a bucket, from the (sign-applied) table entry of te_load_niels: (2 x : 2 y : 2 : 2 x y) with ONE
// multiplication (by 1 / D) instead of a seven-multiplication addition to the neutral element.  x < 5 p, y < 2 p, normalised.
__device__ __forceinline__ TEU teu_from_niels(const FqU& ym, const FqU& yp, const FqU& k2) {
    constexpr u32 idm[14] = TE_INV_D_U;
    TEU a;
    a.x = fqu_normalize(fqu_sub_lazy<4>(yp, ym));
    a.y = fqu_normalize(fqu_add_lazy(yp, ym));
    a.z = fqu_normalize(fqu_add_lazy(fqu_one(), fqu_one()));
    a.t = fqu_mul(k2, fqu_const(idm));
    return a;
}

}  // namespace czk

// field.h -- BLS12-377 Fr / Fq / Fq2 arithmetic for gfx950 (and the host side of the same library).
//
// Values live in HBM exactly as the reference stores them: little-endian u64 limbs in Montgomery form
// with R = 2^256 (Fr) / 2^384 (Fq), always fully reduced (algebra/ff/src/fields/macros.rs:103-108,237-246).
// In registers an element is 2*N64 32-bit limbs, because CDNA4's integer multiplier is 32x32
// (v_mad_u64_u32); the Montgomery radix is unchanged, so every result is the same unique reduced
// representative the reference computes with 64-bit limbs (fields/arithmetic.rs:7-56) -- bit-exact.
//
// Device multiply = product-scanning (Comba/FIPS) Montgomery: one 96-bit column accumulator, each
// partial product is ONE v_mad_u64_u32 (64-bit accumulate) + ONE v_addc_co_u32 (carry into the third
// word).  Both moduli are == 1 mod 2^32, so -p^-1 mod 2^32 = 0xffffffff: the Montgomery quotient digit
// is m = -acc0 and m*p[0] = m costs an add instead of a multiply.  No MFMA: there is no dense
// contraction anywhere on this path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CZK_HD __host__ __device__ __forceinline__
#define CZK_D __device__ __forceinline__
#else
#define CZK_HD inline
#define CZK_D inline
#endif

// Setup / reduction kernels are built with -DCZK_NOINLINE_MUL: the Montgomery multiply becomes a real
// function (one copy per translation unit) instead of ~700 instructions inlined at every call site,
// which keeps those kernels' code size and hipcc's compile time bounded.  Hot kernels (NTT butterflies,
// bucket accumulation) are built without it.
#if defined(CZK_NOINLINE_MUL) && defined(__HIPCC__)
#define CZK_MUL_ATTR __host__ __device__ __attribute__((noinline))
#else
#define CZK_MUL_ATTR CZK_HD
#endif

namespace czk {

typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// Parameters (32-bit limb views of the constants in curves/bls12_377/src/fields/{fr,fq}.rs)
// ---------------------------------------------------------------------------------------------
struct FrParams {
    static constexpr int N = 8;          // 32-bit limbs
    static constexpr int BITS = 253;     // fr.rs:42 MODULUS_BITS
    // fr.rs:33-40 MODULUS
    static CZK_HD u32 p(int i) {
        constexpr u32 m[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu,
                              0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
        return m[i];
    }
    // fr.rs:48-53 R = 2^256 mod r  (Montgomery one)
    static CZK_HD u32 r(int i) {
        constexpr u32 m[8] = {0xfffffff3u, 0x7d1c7fffu, 0x6ffffff2u, 0x7257f50fu,
                              0x512c0feeu, 0x16d81575u, 0x2bbb9a9du, 0x0d4bda32u};
        return m[i];
    }
    // fr.rs:55-61 R2 = R^2 mod r
    static CZK_HD u32 r2(int i) {
        constexpr u32 m[8] = {0xb861857bu, 0x25d577bau, 0x8860591fu, 0xcc2c27b5u,
                              0xe5dc8593u, 0xa7cc008fu, 0xeff1c939u, 0x011fdae7u};
        return m[i];
    }
};

struct FqParams {
    static constexpr int N = 12;
    static constexpr int BITS = 377;     // fq.rs:37
    // fq.rs:26-35 MODULUS
    static CZK_HD u32 p(int i) {
        constexpr u32 m[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                               0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
        return m[i];
    }
    // fq.rs:43-50 R = 2^384 mod q
    static CZK_HD u32 r(int i) {
        constexpr u32 m[12] = {0xffffff68u, 0x02cdffffu, 0x7fffffb1u, 0x51409f83u, 0x8a7d3ff2u, 0x9f7db3a9u,
                               0x6e7c6305u, 0x7b4e97b7u, 0x803c84e8u, 0x4cf495bfu, 0xe2fdf49au, 0x008d6661u};
        return m[i];
    }
    // fq.rs:52-60 R2
    static CZK_HD u32 r2(int i) {
        constexpr u32 m[12] = {0x9400cd22u, 0xb786686cu, 0xb00431b1u, 0x0329fcaau, 0x62d6b46du, 0x22a5f111u,
                               0x827dc3acu, 0xbfdf7d03u, 0x41790bf9u, 0x837e92f0u, 0x1e914b88u, 0x006dfccbu};
        return m[i];
    }
};

// ---------------------------------------------------------------------------------------------
// Fp<P>: prime field element, 32-bit limbs
// ---------------------------------------------------------------------------------------------
template <class P>
struct alignas(16) Fp {
    static constexpr int N = P::N;
    u32 l[N];

    static CZK_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    static CZK_HD Fp one() {   // Montgomery one = R (macros.rs:263-265)
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::r(i);
        return r;
    }
    static CZK_HD Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::r2(i);
        return r;
    }
    CZK_HD bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    CZK_HD bool operator==(const Fp& b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    CZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - p if a >= p else a   (macros.rs:237-246 reduce)
template <class P>
CZK_HD void fp_reduce(Fp<P>& a) {
    constexpr int N = P::N;
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 t = (u64)a.l[i] - P::p(i) - borrow;
        d[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < N; i++) a.l[i] = d[i];
    }
}

// macros.rs:663-669 add_assign.  (a + b < 2p < 2^(32N): never carries out of the top limb.)
template <class P>
CZK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 t = (u64)a.l[i] + b.l[i] + c;
        r.l[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    fp_reduce(r);
    return r;
}

// macros.rs:672-680 sub_assign: a - b, plus p when that borrows.
template <class P>
CZK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 t = (u64)a.l[i] - b.l[i] - borrow;
        r.l[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
    u32 mask = 0u - borrow;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 t = (u64)r.l[i] + (P::p(i) & mask) + c;
        r.l[i] = (u32)t;
        c = (u32)(t >> 32);
    }
    return r;
}

// macros.rs:297-304 double_in_place
template <class P>
CZK_HD Fp<P> fp_dbl(const Fp<P>& a) {
    constexpr int N = P::N;
    Fp<P> r;
    u32 top = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        r.l[i] = (a.l[i] << 1) | top;
        top = a.l[i] >> 31;
    }
    fp_reduce(r);
    return r;
}

// macros.rs:605-617 neg
template <class P>
CZK_HD Fp<P> fp_neg(const Fp<P>& a) {
    constexpr int N = P::N;
    Fp<P> r;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    u32 mask = nz ? 0xffffffffu : 0u;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 t = (u64)(P::p(i) & mask) - a.l[i] - borrow;
        r.l[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
    return r;
}

// ------------------------------------------------------------------------------- multiplication
// 96-bit column accumulator (lo64, hi32).  acc += x*y.
struct Acc96 {
    u64 lo;
    u32 hi;
};

CZK_HD void acc_mad(Acc96& a, u32 x, u32 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one 32x32+64 multiply-add, carry-out into the third word
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(a.lo), "+v"(a.hi)
        : "v"(x), "v"(y)
        : "vcc");
#else
    u64 prod = (u64)x * y;
    u64 s = a.lo + prod;
    a.hi += (s < prod) ? 1u : 0u;
    a.lo = s;
#endif
}
CZK_HD void acc_add32(Acc96& a, u32 x) {
    u64 s = a.lo + x;
    a.hi += (s < (u64)x) ? 1u : 0u;
    a.lo = s;
}
CZK_HD void acc_shift(Acc96& a) {
    a.lo = (a.lo >> 32) | ((u64)a.hi << 32);
    a.hi = 0;
}

// Montgomery product a*b*R^-1 mod p, fully reduced == fields/arithmetic.rs:7-56 (value-identical).
template <class P>
CZK_MUL_ATTR Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    u32 m[N];
    Fp<P> r;
    Acc96 acc = {0, 0};
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i < k; i++) acc_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) acc_mad(acc, m[i], P::p(k - i));
        acc_mad(acc, a.l[k], b.l[0]);
        m[k] = 0u - (u32)acc.lo;          // -p^-1 mod 2^32 == 0xffffffff for both fields
        acc_add32(acc, m[k]);             // + m[k]*p[0], p[0] == 1  -> low word becomes 0
        acc_shift(acc);
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc_mad(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) acc_mad(acc, m[i], P::p(k - i));
        r.l[k - N] = (u32)acc.lo;
        acc_shift(acc);
    }
    r.l[N - 1] = (u32)acc.lo;             // < 2p < 2^(32N): nothing above this word
    fp_reduce(r);
    return r;
}

// square: same value as fp_mul(a, a) (fields/arithmetic.rs:84-170 computes the same residue).  A dedicated
// squaring (N(N+1)/2 products, off-diagonal part doubled per column) was evaluated and is break-even here: a
// product costs only 2 instructions (v_mad_u64_u32 + v_addc), and the per-column doubling/merge costs ~6.
template <class P>
CZK_HD Fp<P> fp_sqr(const Fp<P>& a) {
    return fp_mul(a, a);
}

// fields/arithmetic.rs:59-81 into_repr: Montgomery -> canonical = a * 1 * R^-1
template <class P>
CZK_HD Fp<P> fp_into_repr(const Fp<P>& a) {
    Fp<P> one_plain = Fp<P>::zero();
    one_plain.l[0] = 1;
    return fp_mul(a, one_plain);
}
// macros.rs:443-454 from_repr: canonical -> Montgomery = a * R2 * R^-1
template <class P>
CZK_HD Fp<P> fp_from_repr(const Fp<P>& a) {
    return fp_mul(a, Fp<P>::r2());
}

// a^e, e given as 64-bit words little-endian (Field::pow)
template <class P>
CZK_HD Fp<P> fp_pow(const Fp<P>& a, const u64* e, int words) {
    Fp<P> res = Fp<P>::one();
    bool started = false;
    for (int i = words * 64 - 1; i >= 0; i--) {
        if (started) res = fp_sqr(res);
        if ((e[i / 64] >> (i % 64)) & 1) {
            res = started ? fp_mul(res, a) : a;
            started = true;
        }
    }
    return res;
}
template <class P>
CZK_HD Fp<P> fp_pow_u64(const Fp<P>& a, u64 e) {
    return fp_pow(a, &e, 1);
}

// Inverse by Fermat: a^(p-2).  Value-identical to macros.rs:367-421 (the inverse is unique); used only
// for O(1)-per-call work (domain constants, one Z^-1 per MSM result), never in a hot loop.
template <class P>
CZK_HD Fp<P> fp_inv(const Fp<P>& a) {
    constexpr int N = P::N;
    u64 e[N / 2];
    // p - 2  (p is odd and p[0] == 1 mod 2^32 so the low limb is 0xffffffff after borrow)
    u32 t[N];
    u32 borrow = 2;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u64 d = (u64)P::p(i) - borrow;
        t[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
#pragma unroll
    for (int i = 0; i < N / 2; i++) e[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
    return fp_pow(a, e, N / 2);
}

// memory <-> registers.  HBM layout = the reference's u64 limbs = our u32 limbs on a little-endian machine.
template <class P>
CZK_HD Fp<P> fp_load(const u64* p) {
    Fp<P> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        uint4 v = q[i];
        r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
    }
    return r;
}
template <class P>
CZK_HD void fp_store(u64* p, const Fp<P>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        uint4 v;
        v.x = a.l[4 * i]; v.y = a.l[4 * i + 1]; v.z = a.l[4 * i + 2]; v.w = a.l[4 * i + 3];
        q[i] = v;
    }
}

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// ---------------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 5)   (fields/models/quadratic_extension.rs, curves/bls12_377/src/fields/fq2.rs)
// ---------------------------------------------------------------------------------------------
struct alignas(16) Fq2 {
    Fq c0, c1;
    static CZK_HD Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static CZK_HD Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    CZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    CZK_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
    CZK_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
};

// beta * x with beta = -5  (fq2.rs:29-34)
CZK_HD Fq fq_mul_by_nonresidue(const Fq& x) {
    Fq t = fp_dbl(fp_dbl(x));      // 4x
    return fp_neg(fp_add(t, x));   // -(5x)
}

// overloads so the curve code is generic over the base field
CZK_HD Fq f_add(const Fq& a, const Fq& b) { return fp_add(a, b); }
CZK_HD Fq f_sub(const Fq& a, const Fq& b) { return fp_sub(a, b); }
CZK_HD Fq f_dbl(const Fq& a) { return fp_dbl(a); }
CZK_HD Fq f_neg(const Fq& a) { return fp_neg(a); }
CZK_HD Fq f_mul(const Fq& a, const Fq& b) { return fp_mul(a, b); }
CZK_HD Fq f_sqr(const Fq& a) { return fp_sqr(a); }
CZK_HD Fq f_inv(const Fq& a) { return fp_inv(a); }

CZK_HD Fq2 f_add(const Fq2& a, const Fq2& b) { return Fq2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
CZK_HD Fq2 f_sub(const Fq2& a, const Fq2& b) { return Fq2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
CZK_HD Fq2 f_dbl(const Fq2& a) { return Fq2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
CZK_HD Fq2 f_neg(const Fq2& a) { return Fq2{fp_neg(a.c0), fp_neg(a.c1)}; }
// quadratic_extension.rs:571-583 Karatsuba
CZK_HD Fq2 f_mul(const Fq2& a, const Fq2& b) {
    Fq v0 = fp_mul(a.c0, b.c0);
    Fq v1 = fp_mul(a.c1, b.c1);
    Fq s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    s = fp_sub(fp_sub(s, v0), v1);
    return Fq2{fp_add(v0, fq_mul_by_nonresidue(v1)), s};
}
// quadratic_extension.rs:257-305 (generic-beta branch): c0 = (c0-c1)(c0-beta c1) + (beta+1) c0 c1, c1 = 2 c0 c1
CZK_HD Fq2 f_sqr(const Fq2& a) {
    Fq v0 = fp_sub(a.c0, a.c1);
    Fq v3 = fp_sub(a.c0, fq_mul_by_nonresidue(a.c1));
    Fq v2 = fp_mul(a.c0, a.c1);
    v0 = fp_mul(v0, v3);
    return Fq2{fp_add(fp_add(v0, v2), fq_mul_by_nonresidue(v2)), fp_dbl(v2)};
}
// quadratic_extension.rs:308-324
CZK_HD Fq2 f_inv(const Fq2& a) {
    Fq n = fp_sub(fp_sqr(a.c0), fq_mul_by_nonresidue(fp_sqr(a.c1)));
    Fq ni = fp_inv(n);
    return Fq2{fp_mul(a.c0, ni), fp_neg(fp_mul(a.c1, ni))};
}

template <class F> struct FieldIO;
template <> struct FieldIO<Fq> {
    static constexpr int W64 = 6;
    static CZK_HD Fq load(const u64* p) { return fp_load<FqParams>(p); }
    static CZK_HD void store(u64* p, const Fq& a) { fp_store<FqParams>(p, a); }
};
template <> struct FieldIO<Fq2> {
    static constexpr int W64 = 12;
    static CZK_HD Fq2 load(const u64* p) { return Fq2{fp_load<FqParams>(p), fp_load<FqParams>(p + 6)}; }
    static CZK_HD void store(u64* p, const Fq2& a) {
        fp_store<FqParams>(p, a.c0);
        fp_store<FqParams>(p + 6, a.c1);
    }
};

}  // namespace czk

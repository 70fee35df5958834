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

#include <type_traits>

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

#if defined(__HIP_DEVICE_COMPILE__)
// ---------------------------------------------------------------------------------------------
// Device add / sub / reduce as explicit carry chains (generated for N = 8 and N = 12).  hipcc lowers the
// portable u64 carry emulation below to ~100 instructions per operation (64-bit adds, compares, sign
// extensions); these are 3N: add-with-carry, subtract-with-borrow, select.
// ---------------------------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ Fp<P> fp_add_asm(const Fp<P>& a, const Fp<P>& b, std::integral_constant<int, 8>) {
    Fp<P> r;
    u32 d[8];
    asm("v_add_co_u32 %0, vcc, %16, %24\n\t"
            "v_addc_co_u32 %1, vcc, %17, %25, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %18, %26, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %19, %27, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %20, %28, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %21, %29, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %22, %30, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %23, %31, vcc\n\t"
            "v_sub_co_u32 %8, vcc, %0, %32\n\t"
            "v_subb_co_u32 %9, vcc, %1, %33, vcc\n\t"
            "v_subb_co_u32 %10, vcc, %2, %34, vcc\n\t"
            "v_subb_co_u32 %11, vcc, %3, %35, vcc\n\t"
            "v_subb_co_u32 %12, vcc, %4, %36, vcc\n\t"
            "v_subb_co_u32 %13, vcc, %5, %37, vcc\n\t"
            "v_subb_co_u32 %14, vcc, %6, %38, vcc\n\t"
            "v_subb_co_u32 %15, vcc, %7, %39, vcc\n\t"
            "v_cndmask_b32 %0, %8, %0, vcc\n\t"
            "v_cndmask_b32 %1, %9, %1, vcc\n\t"
            "v_cndmask_b32 %2, %10, %2, vcc\n\t"
            "v_cndmask_b32 %3, %11, %3, vcc\n\t"
            "v_cndmask_b32 %4, %12, %4, vcc\n\t"
            "v_cndmask_b32 %5, %13, %5, vcc\n\t"
            "v_cndmask_b32 %6, %14, %6, vcc\n\t"
            "v_cndmask_b32 %7, %15, %7, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7))
        : "vcc");
    return r;
}
template <class P>
__device__ __forceinline__ Fp<P> fp_sub_asm(const Fp<P>& a, const Fp<P>& b, std::integral_constant<int, 8>) {
    Fp<P> r;
    u32 d[8];
    asm("v_sub_co_u32 %0, vcc, %16, %24\n\t"
            "v_subb_co_u32 %1, vcc, %17, %25, vcc\n\t"
            "v_subb_co_u32 %2, vcc, %18, %26, vcc\n\t"
            "v_subb_co_u32 %3, vcc, %19, %27, vcc\n\t"
            "v_subb_co_u32 %4, vcc, %20, %28, vcc\n\t"
            "v_subb_co_u32 %5, vcc, %21, %29, vcc\n\t"
            "v_subb_co_u32 %6, vcc, %22, %30, vcc\n\t"
            "v_subb_co_u32 %7, vcc, %23, %31, vcc\n\t"
            "v_cndmask_b32 %8, 0, %32, vcc\n\t"
            "v_cndmask_b32 %9, 0, %33, vcc\n\t"
            "v_cndmask_b32 %10, 0, %34, vcc\n\t"
            "v_cndmask_b32 %11, 0, %35, vcc\n\t"
            "v_cndmask_b32 %12, 0, %36, vcc\n\t"
            "v_cndmask_b32 %13, 0, %37, vcc\n\t"
            "v_cndmask_b32 %14, 0, %38, vcc\n\t"
            "v_cndmask_b32 %15, 0, %39, vcc\n\t"
            "v_add_co_u32 %0, vcc, %0, %8\n\t"
            "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %7, %15, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7))
        : "vcc");
    return r;
}
template <class P>
__device__ __forceinline__ void fp_reduce_asm(Fp<P>& r, std::integral_constant<int, 8>) {
    u32 d[8];
    asm("v_sub_co_u32 %8, vcc, %0, %16\n\t"
            "v_subb_co_u32 %9, vcc, %1, %17, vcc\n\t"
            "v_subb_co_u32 %10, vcc, %2, %18, vcc\n\t"
            "v_subb_co_u32 %11, vcc, %3, %19, vcc\n\t"
            "v_subb_co_u32 %12, vcc, %4, %20, vcc\n\t"
            "v_subb_co_u32 %13, vcc, %5, %21, vcc\n\t"
            "v_subb_co_u32 %14, vcc, %6, %22, vcc\n\t"
            "v_subb_co_u32 %15, vcc, %7, %23, vcc\n\t"
            "v_cndmask_b32 %0, %8, %0, vcc\n\t"
            "v_cndmask_b32 %1, %9, %1, vcc\n\t"
            "v_cndmask_b32 %2, %10, %2, vcc\n\t"
            "v_cndmask_b32 %3, %11, %3, vcc\n\t"
            "v_cndmask_b32 %4, %12, %4, vcc\n\t"
            "v_cndmask_b32 %5, %13, %5, vcc\n\t"
            "v_cndmask_b32 %6, %14, %6, vcc\n\t"
            "v_cndmask_b32 %7, %15, %7, vcc"
        : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7))
        : "vcc");
}
template <class P>
__device__ __forceinline__ Fp<P> fp_add_asm(const Fp<P>& a, const Fp<P>& b, std::integral_constant<int, 12>) {
    Fp<P> r;
    u32 d[12];
    asm("v_add_co_u32 %0, vcc, %24, %36\n\t"
            "v_addc_co_u32 %1, vcc, %25, %37, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %26, %38, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %27, %39, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %28, %40, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %29, %41, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %30, %42, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %31, %43, vcc\n\t"
            "v_addc_co_u32 %8, vcc, %32, %44, vcc\n\t"
            "v_addc_co_u32 %9, vcc, %33, %45, vcc\n\t"
            "v_addc_co_u32 %10, vcc, %34, %46, vcc\n\t"
            "v_addc_co_u32 %11, vcc, %35, %47, vcc\n\t"
            "v_sub_co_u32 %12, vcc, %0, %48\n\t"
            "v_subb_co_u32 %13, vcc, %1, %49, vcc\n\t"
            "v_subb_co_u32 %14, vcc, %2, %50, vcc\n\t"
            "v_subb_co_u32 %15, vcc, %3, %51, vcc\n\t"
            "v_subb_co_u32 %16, vcc, %4, %52, vcc\n\t"
            "v_subb_co_u32 %17, vcc, %5, %53, vcc\n\t"
            "v_subb_co_u32 %18, vcc, %6, %54, vcc\n\t"
            "v_subb_co_u32 %19, vcc, %7, %55, vcc\n\t"
            "v_subb_co_u32 %20, vcc, %8, %56, vcc\n\t"
            "v_subb_co_u32 %21, vcc, %9, %57, vcc\n\t"
            "v_subb_co_u32 %22, vcc, %10, %58, vcc\n\t"
            "v_subb_co_u32 %23, vcc, %11, %59, vcc\n\t"
            "v_cndmask_b32 %0, %12, %0, vcc\n\t"
            "v_cndmask_b32 %1, %13, %1, vcc\n\t"
            "v_cndmask_b32 %2, %14, %2, vcc\n\t"
            "v_cndmask_b32 %3, %15, %3, vcc\n\t"
            "v_cndmask_b32 %4, %16, %4, vcc\n\t"
            "v_cndmask_b32 %5, %17, %5, vcc\n\t"
            "v_cndmask_b32 %6, %18, %6, vcc\n\t"
            "v_cndmask_b32 %7, %19, %7, vcc\n\t"
            "v_cndmask_b32 %8, %20, %8, vcc\n\t"
            "v_cndmask_b32 %9, %21, %9, vcc\n\t"
            "v_cndmask_b32 %10, %22, %10, vcc\n\t"
            "v_cndmask_b32 %11, %23, %11, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(r.l[8]), "=&v"(r.l[9]), "=&v"(r.l[10]), "=&v"(r.l[11]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(a.l[9]), "v"(a.l[10]), "v"(a.l[11]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8]), "v"(b.l[9]), "v"(b.l[10]), "v"(b.l[11]), "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7)), "v"(P::p(8)), "v"(P::p(9)), "v"(P::p(10)), "v"(P::p(11))
        : "vcc");
    return r;
}
template <class P>
__device__ __forceinline__ Fp<P> fp_sub_asm(const Fp<P>& a, const Fp<P>& b, std::integral_constant<int, 12>) {
    Fp<P> r;
    u32 d[12];
    asm("v_sub_co_u32 %0, vcc, %24, %36\n\t"
            "v_subb_co_u32 %1, vcc, %25, %37, vcc\n\t"
            "v_subb_co_u32 %2, vcc, %26, %38, vcc\n\t"
            "v_subb_co_u32 %3, vcc, %27, %39, vcc\n\t"
            "v_subb_co_u32 %4, vcc, %28, %40, vcc\n\t"
            "v_subb_co_u32 %5, vcc, %29, %41, vcc\n\t"
            "v_subb_co_u32 %6, vcc, %30, %42, vcc\n\t"
            "v_subb_co_u32 %7, vcc, %31, %43, vcc\n\t"
            "v_subb_co_u32 %8, vcc, %32, %44, vcc\n\t"
            "v_subb_co_u32 %9, vcc, %33, %45, vcc\n\t"
            "v_subb_co_u32 %10, vcc, %34, %46, vcc\n\t"
            "v_subb_co_u32 %11, vcc, %35, %47, vcc\n\t"
            "v_cndmask_b32 %12, 0, %48, vcc\n\t"
            "v_cndmask_b32 %13, 0, %49, vcc\n\t"
            "v_cndmask_b32 %14, 0, %50, vcc\n\t"
            "v_cndmask_b32 %15, 0, %51, vcc\n\t"
            "v_cndmask_b32 %16, 0, %52, vcc\n\t"
            "v_cndmask_b32 %17, 0, %53, vcc\n\t"
            "v_cndmask_b32 %18, 0, %54, vcc\n\t"
            "v_cndmask_b32 %19, 0, %55, vcc\n\t"
            "v_cndmask_b32 %20, 0, %56, vcc\n\t"
            "v_cndmask_b32 %21, 0, %57, vcc\n\t"
            "v_cndmask_b32 %22, 0, %58, vcc\n\t"
            "v_cndmask_b32 %23, 0, %59, vcc\n\t"
            "v_add_co_u32 %0, vcc, %0, %12\n\t"
            "v_addc_co_u32 %1, vcc, %1, %13, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %2, %14, vcc\n\t"
            "v_addc_co_u32 %3, vcc, %3, %15, vcc\n\t"
            "v_addc_co_u32 %4, vcc, %4, %16, vcc\n\t"
            "v_addc_co_u32 %5, vcc, %5, %17, vcc\n\t"
            "v_addc_co_u32 %6, vcc, %6, %18, vcc\n\t"
            "v_addc_co_u32 %7, vcc, %7, %19, vcc\n\t"
            "v_addc_co_u32 %8, vcc, %8, %20, vcc\n\t"
            "v_addc_co_u32 %9, vcc, %9, %21, vcc\n\t"
            "v_addc_co_u32 %10, vcc, %10, %22, vcc\n\t"
            "v_addc_co_u32 %11, vcc, %11, %23, vcc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(r.l[8]), "=&v"(r.l[9]), "=&v"(r.l[10]), "=&v"(r.l[11]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]), "v"(a.l[8]), "v"(a.l[9]), "v"(a.l[10]), "v"(a.l[11]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "v"(b.l[8]), "v"(b.l[9]), "v"(b.l[10]), "v"(b.l[11]), "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7)), "v"(P::p(8)), "v"(P::p(9)), "v"(P::p(10)), "v"(P::p(11))
        : "vcc");
    return r;
}
template <class P>
__device__ __forceinline__ void fp_reduce_asm(Fp<P>& r, std::integral_constant<int, 12>) {
    u32 d[12];
    asm("v_sub_co_u32 %12, vcc, %0, %24\n\t"
            "v_subb_co_u32 %13, vcc, %1, %25, vcc\n\t"
            "v_subb_co_u32 %14, vcc, %2, %26, vcc\n\t"
            "v_subb_co_u32 %15, vcc, %3, %27, vcc\n\t"
            "v_subb_co_u32 %16, vcc, %4, %28, vcc\n\t"
            "v_subb_co_u32 %17, vcc, %5, %29, vcc\n\t"
            "v_subb_co_u32 %18, vcc, %6, %30, vcc\n\t"
            "v_subb_co_u32 %19, vcc, %7, %31, vcc\n\t"
            "v_subb_co_u32 %20, vcc, %8, %32, vcc\n\t"
            "v_subb_co_u32 %21, vcc, %9, %33, vcc\n\t"
            "v_subb_co_u32 %22, vcc, %10, %34, vcc\n\t"
            "v_subb_co_u32 %23, vcc, %11, %35, vcc\n\t"
            "v_cndmask_b32 %0, %12, %0, vcc\n\t"
            "v_cndmask_b32 %1, %13, %1, vcc\n\t"
            "v_cndmask_b32 %2, %14, %2, vcc\n\t"
            "v_cndmask_b32 %3, %15, %3, vcc\n\t"
            "v_cndmask_b32 %4, %16, %4, vcc\n\t"
            "v_cndmask_b32 %5, %17, %5, vcc\n\t"
            "v_cndmask_b32 %6, %18, %6, vcc\n\t"
            "v_cndmask_b32 %7, %19, %7, vcc\n\t"
            "v_cndmask_b32 %8, %20, %8, vcc\n\t"
            "v_cndmask_b32 %9, %21, %9, vcc\n\t"
            "v_cndmask_b32 %10, %22, %10, vcc\n\t"
            "v_cndmask_b32 %11, %23, %11, vcc"
        : "+v"(r.l[0]), "+v"(r.l[1]), "+v"(r.l[2]), "+v"(r.l[3]), "+v"(r.l[4]), "+v"(r.l[5]), "+v"(r.l[6]), "+v"(r.l[7]), "+v"(r.l[8]), "+v"(r.l[9]), "+v"(r.l[10]), "+v"(r.l[11]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
        : "v"(P::p(0)), "v"(P::p(1)), "v"(P::p(2)), "v"(P::p(3)), "v"(P::p(4)), "v"(P::p(5)), "v"(P::p(6)), "v"(P::p(7)), "v"(P::p(8)), "v"(P::p(9)), "v"(P::p(10)), "v"(P::p(11))
        : "vcc");
}
#endif

// r = a - p if a >= p else a   (macros.rs:237-246 reduce)
template <class P>
CZK_HD void fp_reduce(Fp<P>& a) {
    constexpr int N = P::N;
#if defined(__HIP_DEVICE_COMPILE__)
    fp_reduce_asm(a, std::integral_constant<int, N>{});
    return;
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_add_asm(a, b, std::integral_constant<int, N>{});
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_sub_asm(a, b, std::integral_constant<int, N>{});
#endif
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
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_add_asm(a, a, std::integral_constant<int, N>{});   // same value as mul2 + reduce; same 3N instructions
#endif
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

// ---------------------------------------------------------------------------------------------
// acc += sum_t x[t] * y[t] for CNT products in ONE asm statement.  hipcc pads every asm statement with an
// s_nop (it cannot see inside), so one statement per product costs 3 issue slots per product instead of 2;
// a whole column per statement brings that back to 2 + 1/CNT.  (generated: CNT = 1..12)
// ---------------------------------------------------------------------------------------------
template <int CNT>
struct MadN;
template <>
struct MadN<1> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0])
            : "vcc");
#else
        for (int t = 0; t < 1; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<2> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1])
            : "vcc");
#else
        for (int t = 0; t < 2; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<3> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2])
            : "vcc");
#else
        for (int t = 0; t < 3; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<4> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3])
            : "vcc");
#else
        for (int t = 0; t < 4; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<5> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4])
            : "vcc");
#else
        for (int t = 0; t < 5; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<6> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5])
            : "vcc");
#else
        for (int t = 0; t < 6; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<7> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6])
            : "vcc");
#else
        for (int t = 0; t < 7; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<8> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7])
            : "vcc");
#else
        for (int t = 0; t < 8; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<9> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8])
            : "vcc");
#else
        for (int t = 0; t < 9; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<10> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %20, %21, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9])
            : "vcc");
#else
        for (int t = 0; t < 10; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<11> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %20, %21, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %22, %23, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10])
            : "vcc");
#else
        for (int t = 0; t < 11; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};
template <>
struct MadN<12> {
    static CZK_HD void run(Acc96& a, const u32* x, const u32* y) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm(
            "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %20, %21, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %22, %23, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "v_mad_u64_u32 %0, vcc, %24, %25, %0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(a.lo), "+v"(a.hi)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11])
            : "vcc");
#else
        for (int t = 0; t < 12; t++) acc_mad(a, x[t], y[t]);
#endif
    }
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
CZK_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Montgomery product a*b*R^-1 mod p, fully reduced == fields/arithmetic.rs:7-56 (value-identical).
// Column k of the product-scanning schedule: sum_i a[i] b[k-i]  +  sum_i m[i] p[k-i]  (m[k] = -acc0 for k < N).
template <class P>
CZK_MUL_ATTR Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    u32 m[N];
    Fp<P> r;
    Acc96 acc = {0, 0};
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;         // products a[i] b[k-i], i = i0 .. min(k, N-1)
        {
            u32 xs[cab], ys[cab];
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = a.l[i0 + t];
                ys[t] = b.l[k - i0 - t];
            }
            MadN<cab>::run(acc, xs, ys);
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;      // products m[i] p[k-i], p index >= 1
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m[i0 + t];
                ys[t] = P::p(k - i0 - t);
            }
            MadN<cmp>::run(acc, xs, ys);
        }
        if constexpr (k < N) {
            m[k] = 0u - (u32)acc.lo;          // -p^-1 mod 2^32 == 0xffffffff for both fields
            acc_add32(acc, m[k]);             // + m[k]*p[0], p[0] == 1  -> low word becomes 0
        } else {
            r.l[k - N] = (u32)acc.lo;
        }
        acc_shift(acc);
    });
    r.l[N - 1] = (u32)acc.lo;                 // < 2p < 2^(32N): nothing above this word
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

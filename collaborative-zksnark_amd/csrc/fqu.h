// fqu.h -- Fq in an UNSATURATED representation for the MSM's hottest loop (device only).
//
// 14 limbs of 28 bits (392 bits), Montgomery radix R' = 2^392.  A column of the product-scanning multiply sums at
// most 14 + 13 products of 2 x 30-bit limbs plus a carry: < 2^64, so the column accumulator is a plain u64 and every
// partial product is ONE v_mad_u64_u32 (the saturated form in field.h needs a v_addc per product to track the
// 65th..96th bits).  p == 1 mod 2^28, so the quotient digit is again m = -acc mod 2^28.  392 - 377 = 15 spare bits
// let additions and subtractions stay LAZY: a multiply accepts operands up to 2^7 p with limbs < 2^30 and returns
// a value < 1.01 p with normalised limbs (no final subtraction), so only the two stored accumulator coordinates
// are re-normalised per mixed addition.  ~486 instructions per multiply against ~620.
//
// Values: a field element a is held as a * R' mod p (+ a multiple of p when lazy).  Bases are converted once at
// registration (multiply by R' mod p in the saturated form, stored as canonical 12 x u32 integers); bucket results
// are converted back with one saturated multiply by 2^376 = R^2 / R', so everything outside the accumulate kernel
// keeps the reference's Montgomery form.  Group elements are unchanged; only the residue system differs.
#pragma once
#include "curve.h"

namespace czk {

struct FqU {
    u32 l[14];
};
constexpr u32 FQU_MASK = (1u << 28) - 1u;

__device__ __forceinline__ u32 fqu_p(int i) {
    constexpr u32 m[14] = {0x00000001u, 0x008c0000u, 0x00000085u, 0x05d44300u, 0x0800170bu, 0x02fba094u, 0x0f1ef362u,
                           0x000f5138u, 0x0a22d9f3u, 0x0a1493b1u, 0x0b05c06cu, 0x010eac63u, 0x0a4617c5u, 0x00001ae3u};
    return m[i];
}
// K p in redundant limb form: every limb but the top is >= U * 2^28, so a_i + L_i - b_i (- ...) never goes negative
// for subtrahend limbs < U * 2^28  (generated; sum L_i 2^(28 i) == K p exactly)
__device__ __forceinline__ u32 fqu_4p(int i) {
    constexpr u32 m[14] = {0x10000004u, 0x122fffffu, 0x10000213u, 0x17510bffu, 0x10005c2cu, 0x1bee8251u, 0x1c7bcd87u,
                           0x103d44e2u, 0x188b67cbu, 0x18524ec5u, 0x1c1701b1u, 0x143ab18du, 0x19185f13u, 0x00006b8du};
    return m[i];
}
__device__ __forceinline__ u32 fqu_8p(int i) {
    constexpr u32 m[14] = {0x10000008u, 0x145fffffu, 0x10000427u, 0x1ea217ffu, 0x1000b859u, 0x17dd04a3u, 0x18f79b10u,
                           0x107a89c6u, 0x1116cf97u, 0x10a49d8cu, 0x182e0364u, 0x1875631cu, 0x1230be27u, 0x0000d71cu};
    return m[i];
}
__device__ __forceinline__ u32 fqu_16p(int i) {
    constexpr u32 m[14] = {0x10000010u, 0x18bfffffu, 0x1000084fu, 0x1d442fffu, 0x100170b4u, 0x1fba0947u, 0x11ef3621u,
                           0x10f5138eu, 0x122d9f2fu, 0x11493b19u, 0x105c06c9u, 0x10eac63au, 0x14617c50u, 0x0001ae39u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_8p_wide(int i) {   // limbs >= 3 * 2^28: absorbs three normalised subtrahends
    constexpr u32 m[14] = {0x30000008u, 0x345ffffdu, 0x30000425u, 0x3ea217fdu, 0x3000b857u, 0x37dd04a1u, 0x38f79b0eu,
                           0x307a89c4u, 0x3116cf95u, 0x30a49d8au, 0x382e0362u, 0x3875631au, 0x3230be25u, 0x0000d71au};
    return m[i];
}
__device__ __forceinline__ FqU fqu_one() {   // R' mod p
    constexpr u32 m[14] = {0x0fff67acu, 0x020fffffu, 0x0fb0d727u, 0x0e9203ffu, 0x0249b0e4u, 0x0e172345u, 0x0955d771u,
                           0x02bf89aau, 0x0b2833b2u, 0x098e116bu, 0x07dc5c97u, 0x00d43e93u, 0x02e3314bu, 0x000003b4u};
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = m[i];
    return r;
}
// saturated-form constants for the conversions (32-bit limbs)
__device__ __forceinline__ Fq fqu_k_to_u() {   // R' mod p
    constexpr u32 m[12] = {0xffff67acu, 0x2720ffffu, 0x3fffb0d7u, 0xb0e4e920u, 0x72345249u, 0x55d771e1u,
                           0x2bf89aa9u, 0xbb2833b2u, 0x9798e116u, 0xe937dc5cu, 0x314b0d43u, 0x003b42e3u};
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = m[i];
    return r;
}
__device__ __forceinline__ Fq fqu_k_from_u() {   // R^2 / R' = 2^376
    Fq r = Fq::zero();
    r.l[11] = 0x01000000u;
    return r;
}

// acc += sum_t x[t] * y[t], one v_mad_u64_u32 per product, one asm statement per column (generated: CNT = 1..14)
template <int CNT>
struct MadU;
template <>
struct MadU<1> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0])
            : "vcc");
    }
};
template <>
struct MadU<2> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1])
            : "vcc");
    }
};
template <>
struct MadU<3> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2])
            : "vcc");
    }
};
template <>
struct MadU<4> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3])
            : "vcc");
    }
};
template <>
struct MadU<5> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4])
            : "vcc");
    }
};
template <>
struct MadU<6> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5])
            : "vcc");
    }
};
template <>
struct MadU<7> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6])
            : "vcc");
    }
};
template <>
struct MadU<8> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7])
            : "vcc");
    }
};
template <>
struct MadU<9> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8])
            : "vcc");
    }
};
template <>
struct MadU<10> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9])
            : "vcc");
    }
};
template <>
struct MadU<11> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10])
            : "vcc");
    }
};
template <>
struct MadU<12> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11])
            : "vcc");
    }
};
template <>
struct MadU<13> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %25, %26, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]), "v"(x[12]), "v"(y[12])
            : "vcc");
    }
};
template <>
struct MadU<14> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %25, %26, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %27, %28, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]), "v"(x[12]), "v"(y[12]), "v"(x[13]), "v"(y[13])
            : "vcc");
    }
};
template <>
struct MadU<0> {
    static __device__ __forceinline__ void run(u64&, const u32*, const u32*) {}
};
// first product of a multiply: acc = x * y (the zero addend is the instruction's inline constant: no register pair to clear)
__device__ __forceinline__ u64 mad_first(u32 x, u32 y) {
    u64 a;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a) : "v"(x), "v"(y) : "vcc");
    return a;
}
// Same with the second operand of every product in a SCALAR register (VOP3 takes one SGPR per instruction): for products with the
// modulus, whose limbs are wave-uniform constants -- they then occupy no vector registers and are never re-materialised with v_mov.
template <int CNT>
struct MadUS;
template <>
struct MadUS<1> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0])
            : "vcc");
    }
};
template <>
struct MadUS<2> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1])
            : "vcc");
    }
};
template <>
struct MadUS<3> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2])
            : "vcc");
    }
};
template <>
struct MadUS<4> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3])
            : "vcc");
    }
};
template <>
struct MadUS<5> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4])
            : "vcc");
    }
};
template <>
struct MadUS<6> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5])
            : "vcc");
    }
};
template <>
struct MadUS<7> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6])
            : "vcc");
    }
};
template <>
struct MadUS<8> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7])
            : "vcc");
    }
};
template <>
struct MadUS<9> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8])
            : "vcc");
    }
};
template <>
struct MadUS<10> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9])
            : "vcc");
    }
};
template <>
struct MadUS<11> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10])
            : "vcc");
    }
};
template <>
struct MadUS<12> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11])
            : "vcc");
    }
};
template <>
struct MadUS<13> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %25, %26, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11]), "v"(x[12]), "s"(y[12])
            : "vcc");
    }
};
template <>
struct MadUS<14> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %25, %26, %0\n\t"
            "v_mad_u64_u32 %0, vcc, %27, %28, %0"
            : "+v"(a)
            : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]), "v"(x[9]), "s"(y[9]), "v"(x[10]), "s"(y[10]), "v"(x[11]), "s"(y[11]), "v"(x[12]), "s"(y[12]), "v"(x[13]), "s"(y[13])
            : "vcc");
    }
};

// a * b / R' mod p (+ possibly p): operands may be lazy (limbs < 2^30, value < 2^7 p); result limbs normalised
// (< 2^28, top limb small), value < 1.01 p.
template <bool HI>
__device__ __forceinline__ FqU fqu_mul_impl(const FqU& a, const FqU& b, const FqU& e) {
    constexpr int N = 14;
    u32 m[N];
    FqU r;
    u64 acc;   // set by the first product (mad_first)
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;
        {
            u32 xs[cab], ys[cab];
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = a.l[i0 + t];
                ys[t] = b.l[k - i0 - t];
            }
            if constexpr (k == 0) acc = mad_first(xs[0], ys[0]);
            else MadU<cab>::run(acc, xs, ys);
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m[i0 + t];
                ys[t] = fqu_p(k - i0 - t);
            }
            MadUS<cmp>::run(acc, xs, ys);   // p's limbs from scalar registers
        }
        if constexpr (k < N) {
            m[k] = (0u - (u32)acc) & FQU_MASK;   // -p^-1 == -1 mod 2^28
            acc += FQU_MASK;                     // == (acc + m[k] * p[0]) after the shift below: carries out of the low 28 bits iff they are non-zero
                                                 // (and needs no zero-extended copy of m[k]: one v_mov per column less)
        } else {
            if constexpr (HI) acc += e.l[k - N];
            r.l[k - N] = (u32)acc & FQU_MASK;
        }
        acc >>= 28;
    });
    r.l[N - 1] = (u32)acc;
    if constexpr (HI) r.l[N - 1] += e.l[N - 1];
    return r;
}
__device__ __forceinline__ FqU fqu_mul(const FqU& a, const FqU& b) { return fqu_mul_impl<false>(a, b, a); }
__device__ __forceinline__ FqU fqu_mul_hi(const FqU& a, const FqU& b, const FqU& e) { return fqu_mul_impl<true>(a, b, e); }   // a b / R' + e, normalised (see fqu_mul_add_hi)

// a * a / R' mod p: 105 products instead of 196 (off-diagonal terms once, against the doubled operand).  Same operand
// range as fqu_mul(a, a): limbs < 2^30, so the doubled limbs fit 31 bits and a column holds at most
// 7 * 2^61 + 2^60 + 13 * 2^56 < 2^64.
__device__ __forceinline__ FqU fqu_sqr(const FqU& a) {
    constexpr int N = 14;
    u32 m[N], a2[N];
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a.l[i] << 1;
    FqU r;
    u64 acc;   // set by the first product (mad_first)
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int i1 = (k + 1) / 2;            // pairs i < k - i  <=>  i < k / 2 (rounded up)
        constexpr int coff = i1 - i0;
        if constexpr (coff > 0) {
            u32 xs[coff], ys[coff];
#pragma unroll
            for (int t = 0; t < coff; t++) {
                xs[t] = a.l[i0 + t];
                ys[t] = a2[k - i0 - t];
            }
            MadU<coff>::run(acc, xs, ys);
        }
        if constexpr (k % 2 == 0) {
            u32 xs[1] = {a.l[k / 2]}, ys[1] = {a.l[k / 2]};
            if constexpr (k == 0) acc = mad_first(xs[0], ys[0]);
            else MadU<1>::run(acc, xs, ys);
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m[i0 + t];
                ys[t] = fqu_p(k - i0 - t);
            }
            MadUS<cmp>::run(acc, xs, ys);   // p's limbs from scalar registers
        }
        if constexpr (k < N) {
            m[k] = (0u - (u32)acc) & FQU_MASK;
            acc += FQU_MASK;                     // see fqu_mul
        } else {
            r.l[k - N] = (u32)acc & FQU_MASK;
        }
        acc >>= 28;
    });
    r.l[N - 1] = (u32)acc;
    return r;
}

// (a * b + c * d) / R' mod p with ONE Montgomery reduction (saves 182 of 756 multiply-adds).  Column capacity: the
// caller guarantees limb(a) * limb(b) < 2^58 and limb(c) * limb(d) < 2^58 (one factor of each product normalised),
// so a column holds < 28 * 2^58 + 13 * 2^56 < 2^63.
template <bool HI>
__device__ __forceinline__ FqU fqu_mul_add_impl(const FqU& a, const FqU& b, const FqU& c, const FqU& d, const FqU& e) {
    constexpr int N = 14;
    u32 m[N];
    FqU r;
    u64 acc;   // set by the first product (mad_first)
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;
        {
            u32 xs[cab], ys[cab];
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = a.l[i0 + t];
                ys[t] = b.l[k - i0 - t];
            }
            if constexpr (k == 0) acc = mad_first(xs[0], ys[0]);
            else MadU<cab>::run(acc, xs, ys);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = c.l[i0 + t];
                ys[t] = d.l[k - i0 - t];
            }
            MadU<cab>::run(acc, xs, ys);
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m[i0 + t];
                ys[t] = fqu_p(k - i0 - t);
            }
            MadUS<cmp>::run(acc, xs, ys);   // p's limbs from scalar registers
        }
        if constexpr (k < N) {
            m[k] = (0u - (u32)acc) & FQU_MASK;
            acc += FQU_MASK;                     // see fqu_mul
        } else {
            if constexpr (HI) acc += e.l[k - N];   // the addend rides the column carries: no separate add + normalise pass afterwards
            r.l[k - N] = (u32)acc & FQU_MASK;
        }
        acc >>= 28;
    });
    r.l[N - 1] = (u32)acc;
    if constexpr (HI) r.l[N - 1] += e.l[N - 1];
    return r;
}
__device__ __forceinline__ FqU fqu_mul_add(const FqU& a, const FqU& b, const FqU& c, const FqU& d) { return fqu_mul_add_impl<false>(a, b, c, d, a); }
// (a b + c d) / R' + e with the limbs of e (lazy, < 2^32, any value that keeps the result below 2^392) added into the HIGH columns before they are
// carried out: the result is the NORMALISED form of mont(a b + c d) + e.  Replaces "multiply, then a limb-wise add / subtract, then a carry pass"
// (68 instructions per Fq) by 14 limb subtractions that form e and 14 column adds.
__device__ __forceinline__ FqU fqu_mul_add_hi(const FqU& a, const FqU& b, const FqU& c, const FqU& d, const FqU& e) { return fqu_mul_add_impl<true>(a, b, c, d, e); }

// (a b + c d + e f + g h) / R' mod p with ONE Montgomery reduction.  Column capacity: every limb product < 2^58
// (at most one lazy factor, < 2^30, per product), so a column holds < 56 * 2^58 + 13 * 2^56 < 2^64.
__device__ __forceinline__ FqU fqu_mul_add4(const FqU& a, const FqU& b, const FqU& c, const FqU& d, const FqU& e, const FqU& f, const FqU& g,
                                            const FqU& h) {
    constexpr int N = 14;
    u32 m[N];
    FqU r;
    u64 acc;   // set by the first product (mad_first)
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;
        {
            u32 xs[cab], ys[cab];
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = a.l[i0 + t];
                ys[t] = b.l[k - i0 - t];
            }
            if constexpr (k == 0) acc = mad_first(xs[0], ys[0]);
            else MadU<cab>::run(acc, xs, ys);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = c.l[i0 + t];
                ys[t] = d.l[k - i0 - t];
            }
            MadU<cab>::run(acc, xs, ys);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = e.l[i0 + t];
                ys[t] = f.l[k - i0 - t];
            }
            MadU<cab>::run(acc, xs, ys);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = g.l[i0 + t];
                ys[t] = h.l[k - i0 - t];
            }
            MadU<cab>::run(acc, xs, ys);
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m[i0 + t];
                ys[t] = fqu_p(k - i0 - t);
            }
            MadUS<cmp>::run(acc, xs, ys);   // p's limbs from scalar registers
        }
        if constexpr (k < N) {
            m[k] = (0u - (u32)acc) & FQU_MASK;
            acc += FQU_MASK;                     // see fqu_mul
        } else {
            r.l[k - N] = (u32)acc & FQU_MASK;
        }
        acc >>= 28;
    });
    r.l[N - 1] = (u32)acc;
    return r;
}

// carry-propagate: limbs < 2^28 afterwards (top limb takes the rest)
__device__ __forceinline__ FqU fqu_normalize(const FqU& a) {
    FqU r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        u32 t = a.l[i] + c;
        r.l[i] = t & FQU_MASK;
        c = t >> 28;
    }
    r.l[13] = a.l[13] + c;
    return r;
}
// a - b + K p, lazy result (limbs < 2^30): a lazy-or-normalised (< 2^29.x), b normalised
template <int K>
__device__ __forceinline__ FqU fqu_sub_lazy(const FqU& a, const FqU& b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        u32 L = K == 4 ? fqu_4p(i) : K == 8 ? fqu_8p(i) : fqu_16p(i);
        r.l[i] = a.l[i] + (L - b.l[i]);
    }
    return r;
}
// a - b - 2 c + 8 p, normalised: a, b, c normalised
__device__ __forceinline__ FqU fqu_sub3_norm(const FqU& a, const FqU& b, const FqU& c) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fqu_8p_wide(i) - b.l[i] - c.l[i] - c.l[i]);
    return fqu_normalize(r);
}

// 12 x 32-bit canonical integer -> 14 x 28-bit limbs
__device__ __forceinline__ FqU fqu_unpack(const Fq& s) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, w = bit >> 5, off = bit & 31;
        u32 lo = w < 12 ? s.l[w] : 0u, hi = (w + 1) < 12 ? s.l[w + 1] : 0u;
        u32 v = off == 0 ? lo : ((lo >> off) | (hi << (32 - off)));
        r.l[i] = i < 13 ? (v & FQU_MASK) : v;
    }
    return r;
}
// normalised 14 x 28-bit limbs (value < 2^384) -> 12 x 32-bit integer
__device__ __forceinline__ Fq fqu_pack(const FqU& a) {
    Fq s;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        // bits [32 w, 32 w + 32) of sum a_i 2^(28 i)
        const int lo_limb = (32 * w) / 28, sh = 32 * w - 28 * lo_limb;
        u32 v = a.l[lo_limb] >> sh;
        int have = 28 - sh;
        int nxt = lo_limb + 1;
        if (have < 32 && nxt < 14) {
            v |= a.l[nxt] << have;
            have += 28;
            nxt++;
        }
        if (have < 32 && nxt < 14) v |= a.l[nxt] << have;
        s.l[w] = v;
    }
    return s;
}

// In-place XYZZ mixed addition in the unsaturated residue system (madd-2008-s; see curve.h xyzz_acc_mixed).
// Accumulator invariants: ax, ay normalised with values < 9.5 p / 5.5 p; azz, azzz multiply outputs.
// Returns false when the exceptional case P == +-Q (H == 0 mod p) may have occurred: the caller then redoes this
// addition in the saturated form.  (H = U2 - X1 + 16 p lies in (6 p, 18 p); it is 0 mod p only if it equals j p,
// and p == 1 mod 2^28 makes the low 28 bits of j p equal j.)
__device__ __forceinline__ bool fqu_xyzz_acc_mixed(FqU& ax, FqU& ay, FqU& azz, FqU& azzz, const FqU& qx, const FqU& qy_lazy) {
    FqU u2 = fqu_mul(qx, azz);
    FqU pp = fqu_sub_lazy<16>(u2, ax);
    if (((pp.l[0] & FQU_MASK) - 6u) <= 12u) return false;
    FqU s2 = fqu_mul(qy_lazy, azzz);
    FqU r = fqu_sub_lazy<8>(s2, ay);
    FqU p2 = fqu_sqr(pp);
    azz = fqu_mul(azz, p2);
    FqU p3 = fqu_mul(pp, p2);
    azzz = fqu_mul(azzz, p3);
    FqU qv = fqu_mul(ax, p2);
    FqU t = fqu_sqr(r);
    ax = fqu_sub3_norm(t, p3, qv);                       // < 1.01 p + 8 p
    FqU d = fqu_normalize(fqu_sub_lazy<16>(qv, ax));     // Q - X3 + 16 p
    FqU nay;                                             // 8 p - Y1 (lazy): Y3 = r d + (-Y1) PPP, one reduction
#pragma unroll
    for (int i = 0; i < 14; i++) nay.l[i] = fqu_8p(i) - ay.l[i];
    ay = fqu_mul_add(r, d, nay, p3);                     // normalised multiply output, < 1.01 p
    return true;
}


// ---------------------------------------------------------------------------------------------
// Full XYZZ addition and doubling in the unsaturated residue system: the bucket REDUCTION (msm_acc.h k_reduce_*_u).  One
// reduction performs ~2.3 full additions per bucket (12M + 2S each) -- about 15 % of the accumulation's instruction count when it
// runs in the saturated form (620 instructions per multiply against 486 here).  Buckets then stay in this residue system from the
// accumulate kernel's store to the reduction's last step ("u-form": each coordinate the canonical-width integer of value * R' + k p,
// packed 12 x u32); only the final result is converted to the reference's Montgomery form.
// Value discipline: x < 9.5 p normalised; y, zz, zzz multiply outputs (< 1.01 p, normalised).  The point at infinity is a flag
// (stored as zz == 0 exactly: a finite point's zz is a non-zero residue below 1.01 p, never the integer 0).
// ---------------------------------------------------------------------------------------------
struct XYZZU {
    FqU x, y, zz, zzz;
    bool inf;
};
__device__ __forceinline__ XYZZU xyzzu_zero() {
    XYZZU r;
    r.inf = true;
#pragma unroll
    for (int i = 0; i < 14; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    return r;
}
__device__ __forceinline__ XYZZU xyzzu_load(const u64* p) {
    const Fq x = fp_load<FqParams>(p), y = fp_load<FqParams>(p + 6), zz = fp_load<FqParams>(p + 12), zzz = fp_load<FqParams>(p + 18);
    XYZZU r;
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) any |= zz.l[i];
    r.inf = any == 0;
    r.x = fqu_unpack(x);
    r.y = fqu_unpack(y);
    r.zz = fqu_unpack(zz);
    r.zzz = fqu_unpack(zzz);
    return r;
}
__device__ __forceinline__ void xyzzu_store(u64* p, const XYZZU& a) {
    if (a.inf) {
        const Fq z = Fq::zero();
        fp_store<FqParams>(p, z);
        fp_store<FqParams>(p + 6, z);
        fp_store<FqParams>(p + 12, z);
        fp_store<FqParams>(p + 18, z);
        return;
    }
    fp_store<FqParams>(p, fqu_pack(a.x));
    fp_store<FqParams>(p + 6, fqu_pack(a.y));
    fp_store<FqParams>(p + 12, fqu_pack(a.zz));
    fp_store<FqParams>(p + 18, fqu_pack(a.zzz));
}
// u-form <-> the saturated Montgomery form of curve.h (rare paths and the final result)
__device__ __forceinline__ XYZZ<Fq> xyzzu_to_sat(const XYZZU& a) {
    if (a.inf) return XYZZ<Fq>::zero();
    const Fq kf = fqu_k_from_u();
    return XYZZ<Fq>{fp_mul(fqu_pack(a.x), kf), fp_mul(fqu_pack(a.y), kf), fp_mul(fqu_pack(a.zz), kf), fp_mul(fqu_pack(a.zzz), kf)};
}
__device__ __forceinline__ XYZZU xyzzu_from_sat(const XYZZ<Fq>& s) {
    XYZZU r;
    r.inf = s.is_zero();
    if (r.inf) return xyzzu_zero();
    const Fq kt = fqu_k_to_u();
    r.x = fqu_unpack(fp_mul(s.x, kt));
    r.y = fqu_unpack(fp_mul(s.y, kt));
    r.zz = fqu_unpack(fp_mul(s.zz, kt));
    r.zzz = fqu_unpack(fp_mul(s.zzz, kt));
    return r;
}
// the complete formulas of curve.h (P == Q -> doubling, P == -Q -> infinity): taken when the filter below cannot rule out H == 0
// (by value: taking the accumulator's address would pin it in scratch memory on the fast path as well)
__device__ __noinline__ XYZZU xyzzu_add_slow(XYZZU a, XYZZU b) { return xyzzu_from_sat(xyzz_add(xyzzu_to_sat(a), xyzzu_to_sat(b))); }

// a += b (add-2008-s, 12M + 2S).  H = U2 - U1 + 4 p lies in (2.99 p, 5.01 p): it is 0 mod p only if it equals j p with j in
// 3..5, and p == 1 mod 2^28 makes the low 28 bits of j p equal j -- a one-compare filter; the ~1-in-2^26 suspicious additions (and
// every genuine P == +-Q) go through the saturated complete formulas.
__device__ __forceinline__ void xyzzu_add(XYZZU& a, const XYZZU& b) {
    if (b.inf) return;
    if (a.inf) {
        a = b;
        return;
    }
    const FqU u1 = fqu_mul(a.x, b.zz);
    const FqU u2 = fqu_mul(b.x, a.zz);
    const FqU pp = fqu_sub_lazy<4>(u2, u1);
    if (((pp.l[0] & FQU_MASK) - 3u) <= 2u) {
        a = xyzzu_add_slow(a, b);
        return;
    }
    const FqU s1 = fqu_mul(a.y, b.zzz);
    const FqU s2 = fqu_mul(b.y, a.zzz);
    const FqU r = fqu_sub_lazy<4>(s2, s1);
    const FqU p2 = fqu_sqr(pp);
    const FqU p3 = fqu_mul(pp, p2);
    const FqU qv = fqu_mul(u1, p2);
    a.zz = fqu_mul(fqu_mul(a.zz, b.zz), p2);
    a.zzz = fqu_mul(fqu_mul(a.zzz, b.zzz), p3);
    const FqU t = fqu_sqr(r);
    a.x = fqu_sub3_norm(t, p3, qv);                            // R^2 - PPP - 2 Q + 8 p  < 9.01 p
    const FqU d = fqu_normalize(fqu_sub_lazy<16>(qv, a.x));   // Q - X3 + 16 p
    FqU ns1;                                                   // 8 p - S1 (lazy): Y3 = R d + (-S1) PPP under one reduction
#pragma unroll
    for (int i = 0; i < 14; i++) ns1.l[i] = fqu_8p(i) - s1.l[i];
    a.y = fqu_mul_add(r, d, ns1, p3);
}
// a = 2 a (dbl-2008-s-1, a = 0: 6M + 3S + the fused Y3).  The subgroup has odd order: no point has Y == 0.
__device__ __forceinline__ void xyzzu_double(XYZZU& a) {
    if (a.inf) return;
    FqU u;
#pragma unroll
    for (int i = 0; i < 14; i++) u.l[i] = a.y.l[i] + a.y.l[i];            // U = 2 Y1, limbs < 2^29
    const FqU v = fqu_sqr(u);
    const FqU w = fqu_mul(u, v);
    const FqU s = fqu_mul(a.x, v);
    const FqU xx = fqu_sqr(a.x);
    FqU m;
#pragma unroll
    for (int i = 0; i < 14; i++) m.l[i] = 3u * xx.l[i];                   // M = 3 X1^2, limbs < 2^30, value < 3.03 p
    const FqU mm = fqu_sqr(m);
    FqU x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = mm.l[i] + (fqu_8p_wide(i) - s.l[i] - s.l[i]);   // M^2 - 2 S + 8 p
    x3 = fqu_normalize(x3);
    const FqU d = fqu_normalize(fqu_sub_lazy<16>(s, x3));                 // S - X3 + 16 p
    FqU nw;
#pragma unroll
    for (int i = 0; i < 14; i++) nw.l[i] = fqu_8p(i) - w.l[i];            // 8 p - W (lazy)
    const FqU y3 = fqu_mul_add(m, d, nw, a.y);                            // M (S - X3) - W Y1
    a.zz = fqu_mul(v, a.zz);
    a.zzz = fqu_mul(w, a.zzz);
    a.x = x3;
    a.y = y3;
}


// ---------------------------------------------------------------------------------------------
// Fq2 over the unsaturated residue system (G2 accumulation).  Discipline: every value is re-normalised after every
// add / sub (limbs < 2^28), so only VALUE bounds need tracking; the lazy-subtraction constants below (K p in
// redundant limb form, generated) were chosen with an interval analysis of the whole mixed addition -- stable
// bounds: X < 85 p, Y < 49 p, H < 145 p, r < 81 p, every multiply output < 26 p, capacity 2^392 = 38968 p.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 fqu_8p_u2(int i) {   // 8 p, every limb but the top >= 2 * 2^28
    constexpr u32 m[14] = {0x20000008u, 0x245ffffeu, 0x20000426u, 0x2ea217feu, 0x2000b858u, 0x27dd04a2u, 0x28f79b0fu,
                           0x207a89c5u, 0x2116cf96u, 0x20a49d8bu, 0x282e0363u, 0x2875631bu, 0x2230be26u, 0x0000d71bu};
    return m[i];
}
__device__ __forceinline__ u32 fqu_16p_u5(int i) {   // 16 p, every limb but the top >= 5 * 2^28
    constexpr u32 m[14] = {0x50000010u, 0x58bffffbu, 0x5000084bu, 0x5d442ffbu, 0x500170b0u, 0x5fba0943u, 0x51ef361du,
                           0x50f5138au, 0x522d9f2bu, 0x51493b15u, 0x505c06c5u, 0x50eac636u, 0x54617c4cu, 0x0001ae35u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_16p_u4(int i) {   // 16 p, every limb but the top >= 4 * 2^28
    constexpr u32 m[14] = {0x40000010u, 0x48bffffcu, 0x4000084cu, 0x4d442ffcu, 0x400170b1u, 0x4fba0944u, 0x41ef361eu,
                           0x40f5138bu, 0x422d9f2cu, 0x41493b16u, 0x405c06c6u, 0x40eac637u, 0x44617c4du, 0x0001ae36u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_256p(int i) {   // 256 p, every limb but the top >= 1 * 2^28
    constexpr u32 m[14] = {0x10000100u, 0x1bffffffu, 0x10008507u, 0x1442ffffu, 0x10170b5cu, 0x1ba0947fu, 0x1ef3622eu,
                           0x1f5138f0u, 0x12d9f2ffu, 0x1493b1a1u, 0x15c06ca0u, 0x1eac63afu, 0x1617c50fu, 0x001ae3a3u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_128p(int i) {   // 128 p, every limb but the top >= 1 * 2^28
    constexpr u32 m[14] = {0x10000080u, 0x15ffffffu, 0x10004283u, 0x1a217fffu, 0x100b85adu, 0x1dd04a3fu, 0x1f79b116u,
                           0x17a89c77u, 0x116cf97fu, 0x1a49d8d0u, 0x12e0364fu, 0x175631d7u, 0x130be287u, 0x000d71d1u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_64p(int i) {   // 64 p, every limb but the top >= 1 * 2^28
    constexpr u32 m[14] = {0x10000040u, 0x12ffffffu, 0x10002141u, 0x1510bfffu, 0x1005c2d6u, 0x1ee8251fu, 0x17bcd88au,
                           0x13d44e3bu, 0x18b67cbfu, 0x1524ec67u, 0x11701b27u, 0x13ab18ebu, 0x1185f143u, 0x0006b8e8u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_64p_u3(int i) {   // 64 p, every limb but the top >= 3 * 2^28
    constexpr u32 m[14] = {0x30000040u, 0x32fffffdu, 0x3000213fu, 0x3510bffdu, 0x3005c2d4u, 0x3ee8251du, 0x37bcd888u,
                           0x33d44e39u, 0x38b67cbdu, 0x3524ec65u, 0x31701b25u, 0x33ab18e9u, 0x3185f141u, 0x0006b8e6u};
    return m[i];
}
__device__ __forceinline__ u32 fqu_32p(int i) {   // 32 p, every limb but the top >= 1 * 2^28
    constexpr u32 m[14] = {0x10000020u, 0x117fffffu, 0x100010a0u, 0x1a885fffu, 0x1002e16au, 0x1f74128fu, 0x13de6c44u,
                           0x11ea271du, 0x145b3e5fu, 0x12927633u, 0x10b80d93u, 0x11d58c75u, 0x18c2f8a1u, 0x00035c73u};
    return m[i];
}

struct Fq2U {
    FqU c0, c1;
};
__device__ __forceinline__ FqU fqu_add_lazy(const FqU& a, const FqU& b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
#define FQU_SUBN(NAME, CONST)                                                        \
    __device__ __forceinline__ FqU NAME(const FqU& a, const FqU& b) {               \
        FqU r;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (CONST(i) - b.l[i]); \
        return fqu_normalize(r);                                                     \
    }
FQU_SUBN(fqu_subn_32, fqu_32p)
FQU_SUBN(fqu_subn_64, fqu_64p)
FQU_SUBN(fqu_subn_128, fqu_128p)
#undef FQU_SUBN

// (a0 + a1 u)(b0 + b1 u), u^2 = -5: Karatsuba (quadratic_extension.rs:571-583), operands normalised
__device__ __forceinline__ Fq2U fq2u_mul(const Fq2U& a, const Fq2U& b) {
    FqU v0 = fqu_mul(a.c0, b.c0);
    FqU v1 = fqu_mul(a.c1, b.c1);
    FqU m = fqu_mul(fqu_add_lazy(a.c0, a.c1), fqu_add_lazy(b.c0, b.c1));
    Fq2U r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.c1.l[i] = m.l[i] + (fqu_8p_u2(i) - v0.l[i] - v1.l[i]);          // m - v0 - v1 + 8p
        r.c0.l[i] = v0.l[i] + (fqu_16p_u5(i) - 5u * v1.l[i]);             // v0 - 5 v1 + 16p
    }
    r.c0 = fqu_normalize(r.c0);
    r.c1 = fqu_normalize(r.c1);
    return r;
}
// (a0 + a1 u)^2: c0 = (a0 - a1)(a0 + 5 a1) - 4 a0 a1, c1 = 2 a0 a1 (quadratic_extension.rs:257-305 with beta = -5)
__device__ __forceinline__ Fq2U fq2u_sqr(const Fq2U& a) {
    FqU d1, d2;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d1.l[i] = a.c0.l[i] + (fqu_256p(i) - a.c1.l[i]);                  // lazy, limbs < 2^30
        d2.l[i] = a.c0.l[i] + 5u * a.c1.l[i];
    }
    d2 = fqu_normalize(d2);
    FqU v = fqu_mul(d1, d2);
    FqU v2 = fqu_mul(a.c0, a.c1);
    Fq2U r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        r.c0.l[i] = v.l[i] + (fqu_16p_u4(i) - 4u * v2.l[i]);
        r.c1.l[i] = v2.l[i] + v2.l[i];
    }
    r.c0 = fqu_normalize(r.c0);
    r.c1 = fqu_normalize(r.c1);
    return r;
}
__device__ __forceinline__ u32 fqu_512p_u5(int i) {   // 512 p, every limb but the top >= 5 * 2^28
    constexpr u32 m[14] = {0x50000200u, 0x57fffffbu, 0x50010a0cu, 0x5885fffbu, 0x502e16b5u, 0x574128fbu, 0x5de6c45au,
                           0x5ea271deu, 0x55b3e5fcu, 0x5927633fu, 0x5b80d93du, 0x5d58c75bu, 0x5c2f8a1cu, 0x0035c743u};
    return m[i];
}
// K p - 5 a, normalised (K = 16: a < 3.2 p; K = 512: a < 102 p): the beta * a1 operand of the schoolbook product below
template <bool BIG>
__device__ __forceinline__ FqU fqu_neg5(const FqU& a) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = (BIG ? fqu_512p_u5(i) : fqu_16p_u5(i)) - 5u * a.l[i];
    return fqu_normalize(r);
}
#ifdef CZK_G2_KARATSUBA   // lab builds only: three limb products instead of four (measured no faster)
#include "lab/fq2u_karatsuba.h"
#endif

// (a0 + a1 u)(b0 + b1 u) with u^2 = -5, schoolbook with ONE Montgomery reduction per component:
//   c0 = a0 b0 + a1 (-5 b1),  c1 = a0 b1 + a1 b0          (4 x 196 + 2 x 182 = 1148 multiply-adds)
// Karatsuba (fq2u_mul) needs 1134 but pays ~190 more instructions re-normalising the combinations of its three products;
// n5b1 = fqu_neg5(b.c1) is shared by every product with the same b.  All operands normalised (limb products < 2^57);
// results are multiply outputs (< 1.01 p + (a0 b0 + a1 n5b1) / 2^392, i.e. < 3 p for the operand values below).
__device__ __forceinline__ Fq2U fq2u_mul_n5(const Fq2U& a, const Fq2U& b, const FqU& n5b1) {
    Fq2U r;
    r.c0 = fqu_mul_add(a.c0, b.c0, a.c1, n5b1);
    r.c1 = fqu_mul_add(a.c0, b.c1, a.c1, b.c0);
    return r;
}
__device__ __forceinline__ bool fqu_low_in(const FqU& a, u32 lo, u32 hi) { return (a.l[0] - lo) <= (hi - lo); }

// In-place XYZZ mixed addition over Fq2U.  Returns false when H == 0 mod p is possible (both components of
// H = U2 - X1 + 128 p equal j p for some j in (43, 145): low normalised limb == j).
__device__ __forceinline__ bool fq2u_xyzz_acc_mixed(Fq2U& ax, Fq2U& ay, Fq2U& azz, Fq2U& azzz, const Fq2U& qx, const Fq2U& qy) {
    // value bounds (units of p): table coordinates < 4, multiply outputs < 3, X < 85, Y < 36, H in (43, 131), r < 67
#ifndef CZK_G2_KARATSUBA   // default: the four-product form with a shared -5 b1 operand (Karatsuba measured no faster: profiles/r03_g2_karatsuba.txt)
    // pp, r and X3 leave their multiplies already combined with their addends (fqu_mul_add_hi / fqu_mul_hi: the addend's limbs enter the high
    // columns of the product and ride its carries), so the three "subtract, then carry-propagate" passes of the straightforward form are gone
    const FqU n5zz = fqu_neg5<false>(azz.c1);
    FqU e0, e1;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        e0.l[i] = fqu_128p(i) - ax.c0.l[i];              // 128 p - X1, limb-wise non-negative (X1 normalised)
        e1.l[i] = fqu_128p(i) - ax.c1.l[i];
    }
    Fq2U pp{fqu_mul_add_hi(qx.c0, azz.c0, qx.c1, n5zz, e0), fqu_mul_add_hi(qx.c0, azz.c1, qx.c1, azz.c0, e1)};   // H = U2 - X1 + 128 p
    if (fqu_low_in(pp.c0, 40, 150) && fqu_low_in(pp.c1, 40, 150)) return false;
    const FqU n5zzz = fqu_neg5<false>(azzz.c1);
    FqU nay0, nay1;                                      // 64 p - Y1 (Y1 < 36 p), lazy: r's addend here, a factor of Y3 below
#pragma unroll
    for (int i = 0; i < 14; i++) {
        nay0.l[i] = fqu_64p(i) - ay.c0.l[i];
        nay1.l[i] = fqu_64p(i) - ay.c1.l[i];
    }
    Fq2U r{fqu_mul_add_hi(qy.c0, azzz.c0, qy.c1, n5zzz, nay0), fqu_mul_add_hi(qy.c0, azzz.c1, qy.c1, azzz.c0, nay1)};   // r = S2 - Y1 + 64 p
    Fq2U p2 = fq2u_sqr(pp);                              // c1 = 2 v2 < 3 p
    const FqU n5p2 = fqu_neg5<false>(p2.c1);
    azz = fq2u_mul_n5(p2, azz, n5zz);
    Fq2U p3 = fq2u_mul_n5(pp, p2, n5p2);
    azzz = fq2u_mul_n5(p3, azzz, n5zzz);
    Fq2U qv = fq2u_mul_n5(ax, p2, n5p2);
    {   // X3 = r^2 - p3 - 2 qv + 64 p with r^2 = ((r0 - r1)(r0 + 5 r1) - 4 r0 r1, 2 r0 r1) (fq2u_sqr): c0 as ONE multiply with everything else as its addend
        FqU d1, d2;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            d1.l[i] = r.c0.l[i] + (fqu_256p(i) - r.c1.l[i]);
            d2.l[i] = r.c0.l[i] + 5u * r.c1.l[i];
        }
        d2 = fqu_normalize(d2);
        const FqU v2 = fqu_mul(r.c0, r.c1);
#pragma unroll
        for (int i = 0; i < 14; i++) {
            e0.l[i] = (fqu_16p_u4(i) - 4u * v2.l[i]) + (fqu_64p_u3(i) - p3.c0.l[i] - qv.c0.l[i] - qv.c0.l[i]);   // < 2^31.3
            e1.l[i] = (v2.l[i] + v2.l[i]) + (fqu_64p_u3(i) - p3.c1.l[i] - qv.c1.l[i] - qv.c1.l[i]);
        }
        ax.c0 = fqu_mul_hi(d1, d2, e0);
        ax.c1 = fqu_normalize(e1);
    }
#else                      // -DCZK_G2_KARATSUBA: Karatsuba products (fq2u_mul_k): 952 instead of 1148 multiply-adds each, outputs < 2.1 p
    Fq2U u2 = fq2u_mul_k(qx, azz);
    Fq2U pp{fqu_subn_128(u2.c0, ax.c0), fqu_subn_128(u2.c1, ax.c1)};
    if (fqu_low_in(pp.c0, 40, 150) && fqu_low_in(pp.c1, 40, 150)) return false;
    Fq2U s2 = fq2u_mul_k(qy, azzz);
    Fq2U r{fqu_subn_64(s2.c0, ay.c0), fqu_subn_64(s2.c1, ay.c1)};
    FqU nay0, nay1;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        nay0.l[i] = fqu_64p(i) - ay.c0.l[i];
        nay1.l[i] = fqu_64p(i) - ay.c1.l[i];
    }
    Fq2U p2 = fq2u_sqr(pp);                              // c1 = 2 v2 < 3 p
    azz = fq2u_mul_k(p2, azz);
    Fq2U p3 = fq2u_mul_k(pp, p2);
    azzz = fq2u_mul_k(p3, azzz);
    Fq2U qv = fq2u_mul_k(ax, p2);
    Fq2U t = fq2u_sqr(r);
#pragma unroll
    for (int i = 0; i < 14; i++) {                                        // X3 = t - p3 - 2 qv + 64 p
        ax.c0.l[i] = t.c0.l[i] + (fqu_64p_u3(i) - p3.c0.l[i] - qv.c0.l[i] - qv.c0.l[i]);
        ax.c1.l[i] = t.c1.l[i] + (fqu_64p_u3(i) - p3.c1.l[i] - qv.c1.l[i] - qv.c1.l[i]);
    }
    ax.c0 = fqu_normalize(ax.c0);
    ax.c1 = fqu_normalize(ax.c1);
#endif
    Fq2U d{fqu_subn_128(qv.c0, ax.c0), fqu_subn_128(qv.c1, ax.c1)};
    // Y3 = d r - Y1 PPP, each component four products under one reduction (operands: d, r, p3 normalised; -Y1 lazy)
    const FqU n5r = fqu_neg5<true>(r.c1);
    const FqU n5p3 = fqu_neg5<false>(p3.c1);
    ay.c0 = fqu_mul_add4(d.c0, r.c0, d.c1, n5r, nay0, p3.c0, nay1, n5p3);
    ay.c1 = fqu_mul_add4(d.c0, r.c1, d.c1, r.c0, nay0, p3.c1, nay1, p3.c0);
    return true;
}

}  // namespace czk

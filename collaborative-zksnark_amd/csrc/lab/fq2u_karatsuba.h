// fq2u_karatsuba.h -- LAB ONLY (-DCZK_G2_KARATSUBA builds of libczk_hip_lab.so): the Karatsuba form of the Fq2 product for the G2 bucket kernel and the
// signed multiply-add helpers it needs.  Built, measured, not adopted (EXPERIMENTS.md, profiles/r03_g2_karatsuba.txt).  Included by fqu.h inside namespace czk.
#pragma once

// signed variant (v_mad_i64_i32): acc += sum_t (int32)x[t] * (int32)y[t] -- for the NEGATED operand of the Karatsuba Fq2 product below
template <int CNT>
struct MadI;
template <>
struct MadI<1> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0])
            : "vcc");
    }
};
template <>
struct MadI<2> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1])
            : "vcc");
    }
};
template <>
struct MadI<3> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2])
            : "vcc");
    }
};
template <>
struct MadI<4> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3])
            : "vcc");
    }
};
template <>
struct MadI<5> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4])
            : "vcc");
    }
};
template <>
struct MadI<6> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5])
            : "vcc");
    }
};
template <>
struct MadI<7> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6])
            : "vcc");
    }
};
template <>
struct MadI<8> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7])
            : "vcc");
    }
};
template <>
struct MadI<9> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8])
            : "vcc");
    }
};
template <>
struct MadI<10> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %19, %20, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9])
            : "vcc");
    }
};
template <>
struct MadI<11> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %21, %22, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10])
            : "vcc");
    }
};
template <>
struct MadI<12> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %23, %24, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11])
            : "vcc");
    }
};
template <>
struct MadI<13> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %25, %26, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]), "v"(x[12]), "v"(y[12])
            : "vcc");
    }
};
template <>
struct MadI<14> {
    static __device__ __forceinline__ void run(u64& a, const u32* x, const u32* y) {
        asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %5, %6, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %7, %8, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %9, %10, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %11, %12, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %13, %14, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %15, %16, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %17, %18, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %19, %20, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %21, %22, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %23, %24, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %25, %26, %0\n\t"
            "v_mad_i64_i32 %0, vcc, %27, %28, %0"
            : "+v"(a)
            : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]), "v"(x[9]), "v"(y[9]), "v"(x[10]), "v"(y[10]), "v"(x[11]), "v"(y[11]), "v"(x[12]), "v"(y[12]), "v"(x[13]), "v"(y[13])
            : "vcc");
    }
};
template <>
struct MadI<0> {
    static __device__ __forceinline__ void run(u64&, const u32*, const u32*) {}
};
__device__ __forceinline__ u64 mad_first_i(u32 x, u32 y) {
    u64 a;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(a) : "v"(x), "v"(y) : "vcc");
    return a;
}

// (a0 + a1 u)(b0 + b1 u) with u^2 = -5: KARATSUBA, column by column, both components under one Montgomery reduction each --
//   V0_k = sum a0_i b0_(k-i),  X_k = sum a1_i b1_(k-i),  M_k = sum (a0 + a1)_i (b0 + b1)_(k-i)      (3 x 196 products)
//   column k of c0 = V0_k - 5 X_k,   column k of c1 = M_k - V0_k - X_k  (>= 0: the identity holds limb-wise)
// = 588 + 2 x 182 = 952 multiply-adds against 1148 for the schoolbook form, for ~6 more 64-bit adds per column.  MEASURED NO FASTER
// (28.56 vs 28.23 ms per 2^20-point 4-lane launch): on gfx950 a VOP3 64-bit add issues at nearly the rate of a v_mad_u64_u32, so the
// 972 combining adds cost what the 1176 saved multiply-adds bought (profiles/r03_g2_karatsuba.txt).  Kept behind -DCZK_G2_KARATSUBA.  -X_k is
// accumulated directly with v_mad_i64_i32 on the negated limbs of a1.  The c0 accumulator is SIGNED (two's complement, arithmetic
// shift between columns): |5 X_k| < 5 * 14 * 2^56 < 2^62.2.  Its value is made non-negative by the constant p R' (p's limbs added
// into columns 14..27), which adds p to the result: c0 < p + 1.01 p + V0 / R' -- still a "multiply output < 3 p" of the analysis in
// fq2u_xyzz_acc_mixed, provided value(a1) value(b1) < p R' / 5 = 7793 p^2 (largest there: 131 p x 3 p).  Operands normalised.
__device__ __forceinline__ Fq2U fq2u_mul_k(const Fq2U& a, const Fq2U& b) {
    constexpr int N = 14;
    u32 sa[N], sb[N], na1[N], m0[N], m1[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        sa[i] = a.c0.l[i] + a.c1.l[i];
        sb[i] = b.c0.l[i] + b.c1.l[i];
        na1[i] = 0u - a.c1.l[i];
    }
    Fq2U r;
    u64 acc0 = 0, acc1 = 0;   // acc0 is read as int64
    static_for<0, 2 * N - 1>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;
        {
            u32 xs[cab], ys[cab];
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = a.c0.l[i0 + t];
                ys[t] = b.c0.l[k - i0 - t];
            }
            u64 v0 = mad_first(xs[0], ys[0]);
            MadU<cab - 1>::run(v0, xs + 1, ys + 1);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = na1[i0 + t];
                ys[t] = b.c1.l[k - i0 - t];
            }
            u64 nx = mad_first_i(xs[0], ys[0]);
            MadI<cab - 1>::run(nx, xs + 1, ys + 1);
#pragma unroll
            for (int t = 0; t < cab; t++) {
                xs[t] = sa[i0 + t];
                ys[t] = sb[k - i0 - t];
            }
            MadU<cab>::run(acc1, xs, ys);
            acc0 += v0 + nx + (nx << 2);
            acc1 += nx - v0;
        }
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[cmp], ys[cmp];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                xs[t] = m0[i0 + t];
                ys[t] = fqu_p(k - i0 - t);
            }
            MadUS<cmp>::run(acc0, xs, ys);
#pragma unroll
            for (int t = 0; t < cmp; t++) xs[t] = m1[i0 + t];
            MadUS<cmp>::run(acc1, xs, ys);
        }
        if constexpr (k < N) {
            m0[k] = (0u - (u32)acc0) & FQU_MASK;
            m1[k] = (0u - (u32)acc1) & FQU_MASK;
            acc0 += FQU_MASK;                     // see fqu_mul
            acc1 += FQU_MASK;
        } else {
            acc0 += fqu_p(k - N);                 // the p R' that keeps c0 non-negative
            r.c0.l[k - N] = (u32)acc0 & FQU_MASK;
            r.c1.l[k - N] = (u32)acc1 & FQU_MASK;
        }
        acc0 = (u64)((long long)acc0 >> 28);
        acc1 >>= 28;
    });
    r.c0.l[N - 1] = (u32)acc0 + fqu_p(N - 1);   // the top limb of the p R' offset
    r.c1.l[N - 1] = (u32)acc1;
    return r;
}


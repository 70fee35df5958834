// fqu_il.h -- K independent Montgomery products of fqu.h run SIDE BY SIDE, multiply-add by multiply-add (device only).
// BUILT, MEASURED, NOT ADOPTED (-DCZK_TE_IL / -DCZK_G2_IL select it in the accumulate kernels).
//
// fqu_mul's column is one chain of v_mad_u64_u32 into one accumulator, and the steps between columns (m = -acc, acc += mask,
// acc >>= 28) are serial too.  The formulas have parallelism -- a twisted Edwards mixed addition is three independent products followed
// by four, an Fq2 product is two -- so fqu_mul_il<K> runs K products with K accumulators, the multiply-adds of a column issued round
// robin (asm blocks of fqu_mad_il.inc): K chains in flight per lane, same values, same instruction count (the generated loop of
// k_accumulate_te has 2 646 multiply-adds, no two consecutive ones on the same accumulator).
// Result (profiles/r03_interleave.txt): isolated 2^20 x 4-lane accumulation 7.839 ms interleaved against 7.836 ms serial (G1), 29.09 against
// 29.30 ms (G2) -- nothing.  Dependent v_mad_u64_u32 issue back to back at full rate on this part (tools/bank_bench.hip: 30.9 T/s at two waves
// per SIMD with ONE chain per lane as with eight; no VGPR-bank effect either), so a serial column costs nothing: there was no latency to hide (the G1 kernel is
// limited by the clock the power management allows, 1.84 GHz; the G2 kernel by its single wave per SIMD -- DESIGN.md section 5).
// In the pipeline the interleaved kernels are SLOWER (83.1 against 77.2 ms per proof): 206 instead of 155 registers per wave (G1), 369
// instead of 304 (G2) leave no room for the NTT / sort / reduction waves that used to run beside the accumulate waves.
#pragma once
#include "fqu.h"

namespace czk {
#include "fqu_mad_il.inc"

template <int K>
struct IlChunk {
    static constexpr int V = K == 2 ? 7 : K == 3 ? 4 : 3;   // terms per asm block, both factors in vector registers
    static constexpr int S = K == 2 ? 7 : K == 3 ? 6 : 5;   // second factor shared, in scalar registers
};
// acc[j] += sum_{t < N} xs[j][OFF + t] * ys[j][OFF + t]
template <int K, int N, int OFF = 0>
__device__ __forceinline__ void mad_il(u64 (&acc)[K], const u32 (&xs)[K][14], const u32 (&ys)[K][14]) {
    if constexpr (N > 0) {
        constexpr int n = N < IlChunk<K>::V ? N : IlChunk<K>::V;
        if constexpr (K == 2) MadILBlock<2, n>::run(acc[0], acc[1], xs[0] + OFF, xs[1] + OFF, ys[0] + OFF, ys[1] + OFF);
        else if constexpr (K == 3) MadILBlock<3, n>::run(acc[0], acc[1], acc[2], xs[0] + OFF, xs[1] + OFF, xs[2] + OFF, ys[0] + OFF, ys[1] + OFF, ys[2] + OFF);
        else MadILBlock<4, n>::run(acc[0], acc[1], acc[2], acc[3], xs[0] + OFF, xs[1] + OFF, xs[2] + OFF, xs[3] + OFF, ys[0] + OFF, ys[1] + OFF, ys[2] + OFF, ys[3] + OFF);
        mad_il<K, N - n, OFF + n>(acc, xs, ys);
    }
}
// acc[j] += sum_{t < N} xs[j][OFF + t] * ps[OFF + t]      (ps uniform: the limbs of p)
template <int K, int N, int OFF = 0>
__device__ __forceinline__ void mad_ils(u64 (&acc)[K], const u32 (&xs)[K][14], const u32 (&ps)[14]) {
    if constexpr (N > 0) {
        constexpr int n = N < IlChunk<K>::S ? N : IlChunk<K>::S;
        if constexpr (K == 2) MadILSBlock<2, n>::run(acc[0], acc[1], xs[0] + OFF, xs[1] + OFF, ps + OFF);
        else if constexpr (K == 3) MadILSBlock<3, n>::run(acc[0], acc[1], acc[2], xs[0] + OFF, xs[1] + OFF, xs[2] + OFF, ps + OFF);
        else MadILSBlock<4, n>::run(acc[0], acc[1], acc[2], acc[3], xs[0] + OFF, xs[1] + OFF, xs[2] + OFF, xs[3] + OFF, ps + OFF);
        mad_ils<K, N - n, OFF + n>(acc, xs, ps);
    }
}

// r[j] = (sum_{q < Q} a[j][q] * b[j][q]) / R' mod p (+ possibly p) for j < K, with ONE Montgomery reduction per j: Q = 1 is fqu_mul
// (operands may be lazy, limbs < 2^30), Q = 2 is fqu_mul_add (one factor of every product normalised).
template <int K, int Q>
__device__ __forceinline__ void fqu_mul_il(const FqU* const (&a)[K][Q], const FqU* const (&b)[K][Q], FqU (&r)[K]) {
    constexpr int N = 14;
    u32 m[K][N];
    u64 acc[K];
    static_for<0, 2 * N - 1>([&](auto KK) {
        constexpr int k = decltype(KK)::value;
        constexpr int i0 = k < N ? 0 : k - N + 1;
        constexpr int cab = (k < N ? k : N - 1) - i0 + 1;
        static_for<0, Q>([&](auto QQ) {
            constexpr int q = decltype(QQ)::value;
            u32 xs[K][N], ys[K][N];
#pragma unroll
            for (int j = 0; j < K; j++)
#pragma unroll
                for (int t = 0; t < cab; t++) {
                    xs[j][t] = a[j][q]->l[i0 + t];
                    ys[j][t] = b[j][q]->l[k - i0 - t];
                }
            if constexpr (k == 0 && q == 0) {
#pragma unroll
                for (int j = 0; j < K; j++) acc[j] = mad_first(xs[j][0], ys[j][0]);
            } else {
                mad_il<K, cab>(acc, xs, ys);
            }
        });
        constexpr int cmp = (k < N ? k - 1 : N - 1) - i0 + 1;
        if constexpr (cmp > 0) {
            u32 xs[K][N], ps[N];
#pragma unroll
            for (int t = 0; t < cmp; t++) {
                ps[t] = fqu_p(k - i0 - t);
#pragma unroll
                for (int j = 0; j < K; j++) xs[j][t] = m[j][i0 + t];
            }
            mad_ils<K, cmp>(acc, xs, ps);
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            if constexpr (k < N) {
                m[j][k] = (0u - (u32)acc[j]) & FQU_MASK;
                acc[j] += FQU_MASK;                  // see fqu_mul
            } else {
                r[j].l[k - N] = (u32)acc[j] & FQU_MASK;
            }
            acc[j] >>= 28;
        }
    });
#pragma unroll
    for (int j = 0; j < K; j++) r[j].l[N - 1] = (u32)acc[j];
}

// ---- G2: the XYZZ mixed addition of fqu.h (fq2u_xyzz_acc_mixed, default schoolbook form) with its products grouped four chains at a
// time: [U2 | S2], [H^2 | R^2], [ZZ' | PPP], [ZZZ' | Q], then Y3's two components.  Same operand forms, bounds and results.
__device__ __forceinline__ void fq2u_mul2_n5_il(const Fq2U& a, const Fq2U& b, const FqU& n5b1, const Fq2U& c, const Fq2U& d, const FqU& n5d1, Fq2U& ab, Fq2U& cd) {
    const FqU* const x[4][2] = {{&a.c0, &a.c1}, {&a.c0, &a.c1}, {&c.c0, &c.c1}, {&c.c0, &c.c1}};
    const FqU* const y[4][2] = {{&b.c0, &n5b1}, {&b.c1, &b.c0}, {&d.c0, &n5d1}, {&d.c1, &d.c0}};
    FqU r[4];
    fqu_mul_il<4, 2>(x, y, r);
    ab = Fq2U{r[0], r[1]};
    cd = Fq2U{r[2], r[3]};
}
// a^2 and b^2 (fq2u_sqr twice: four independent products)
__device__ __forceinline__ void fq2u_sqr2_il(const Fq2U& a, const Fq2U& b, Fq2U& aa, Fq2U& bb) {
    FqU d1a, d2a, d1b, d2b;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        d1a.l[i] = a.c0.l[i] + (fqu_256p(i) - a.c1.l[i]);
        d2a.l[i] = a.c0.l[i] + 5u * a.c1.l[i];
        d1b.l[i] = b.c0.l[i] + (fqu_256p(i) - b.c1.l[i]);
        d2b.l[i] = b.c0.l[i] + 5u * b.c1.l[i];
    }
    d2a = fqu_normalize(d2a);
    d2b = fqu_normalize(d2b);
    const FqU* const x[4][1] = {{&d1a}, {&a.c0}, {&d1b}, {&b.c0}};
    const FqU* const y[4][1] = {{&d2a}, {&a.c1}, {&d2b}, {&b.c1}};
    FqU r[4];
    fqu_mul_il<4, 1>(x, y, r);
#pragma unroll
    for (int i = 0; i < 14; i++) {
        aa.c0.l[i] = r[0].l[i] + (fqu_16p_u4(i) - 4u * r[1].l[i]);
        aa.c1.l[i] = r[1].l[i] + r[1].l[i];
        bb.c0.l[i] = r[2].l[i] + (fqu_16p_u4(i) - 4u * r[3].l[i]);
        bb.c1.l[i] = r[3].l[i] + r[3].l[i];
    }
    aa.c0 = fqu_normalize(aa.c0);
    aa.c1 = fqu_normalize(aa.c1);
    bb.c0 = fqu_normalize(bb.c0);
    bb.c1 = fqu_normalize(bb.c1);
}
__device__ __forceinline__ bool fq2u_xyzz_acc_mixed_il(Fq2U& ax, Fq2U& ay, Fq2U& azz, Fq2U& azzz, const Fq2U& qx, const Fq2U& qy) {
    const FqU n5zz = fqu_neg5<false>(azz.c1), n5zzz = fqu_neg5<false>(azzz.c1);
    Fq2U u2, s2;
    fq2u_mul2_n5_il(qx, azz, n5zz, qy, azzz, n5zzz, u2, s2);
    const Fq2U pp{fqu_subn_128(u2.c0, ax.c0), fqu_subn_128(u2.c1, ax.c1)};
    if (fqu_low_in(pp.c0, 40, 150) && fqu_low_in(pp.c1, 40, 150)) return false;
    const Fq2U r{fqu_subn_64(s2.c0, ay.c0), fqu_subn_64(s2.c1, ay.c1)};
    Fq2U p2, t;
    fq2u_sqr2_il(pp, r, p2, t);
    const FqU n5p2 = fqu_neg5<false>(p2.c1);
    Fq2U p3, zz3;
    fq2u_mul2_n5_il(p2, azz, n5zz, pp, p2, n5p2, zz3, p3);
    Fq2U qv, zzz3;
    fq2u_mul2_n5_il(p3, azzz, n5zzz, ax, p2, n5p2, zzz3, qv);
    azz = zz3;
    azzz = zzz3;
#pragma unroll
    for (int i = 0; i < 14; i++) {                                        // X3 = t - p3 - 2 qv + 64 p
        ax.c0.l[i] = t.c0.l[i] + (fqu_64p_u3(i) - p3.c0.l[i] - qv.c0.l[i] - qv.c0.l[i]);
        ax.c1.l[i] = t.c1.l[i] + (fqu_64p_u3(i) - p3.c1.l[i] - qv.c1.l[i] - qv.c1.l[i]);
    }
    ax.c0 = fqu_normalize(ax.c0);
    ax.c1 = fqu_normalize(ax.c1);
    const Fq2U d{fqu_subn_128(qv.c0, ax.c0), fqu_subn_128(qv.c1, ax.c1)};
    const FqU n5r = fqu_neg5<true>(r.c1), n5p3 = fqu_neg5<false>(p3.c1);
    FqU nay0, nay1;                                                       // 64 p - Y1 (lazy)
#pragma unroll
    for (int i = 0; i < 14; i++) {
        nay0.l[i] = fqu_64p(i) - ay.c0.l[i];
        nay1.l[i] = fqu_64p(i) - ay.c1.l[i];
    }
    const FqU* const x[2][4] = {{&d.c0, &d.c1, &nay0, &nay1}, {&d.c0, &d.c1, &nay0, &nay1}};
    const FqU* const y[2][4] = {{&r.c0, &n5r, &p3.c0, &n5p3}, {&r.c1, &r.c0, &p3.c1, &p3.c0}};
    FqU o[2];
    fqu_mul_il<2, 4>(x, y, o);
    ay = Fq2U{o[0], o[1]};
    return true;
}

}  // namespace czk

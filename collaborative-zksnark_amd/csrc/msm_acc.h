// msm_acc.h -- the MSM's hot kernel: bucket accumulation, one thread per bucket.
// Included by msm_acc_g1.hip (F = Fq) and msm_acc_g2.hip (F = Fq2); those translation units are built with
// the Montgomery multiply inlined.  See msm.hip for the surrounding algorithm.
#pragma once
#include "czk_internal.h"

namespace czk {

// sorted[off .. off+cnt) lists this bucket's points as (w * n_bases + i) | sign<<31; pts holds the window
// multiples 2^(c*w) * P_i in affine Montgomery form.  acc += (+/-) P with madd-2007-bl
// (short_weierstrass_jacobian.rs:570-638, edge cases included).
template <class F>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate(const u64* pts, const u32* sorted, const u32* offsets, const u32* counts,
                                                   const u32* perm, size_t B, size_t sorted_stride, u64* buckets) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const size_t b = perm[(size_t)lane * B + t];   // buckets in descending-population order: equal work per wave
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    Jac<F> acc = Jac<F>::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        Affine<F> p = aff_load<F>(pts + (size_t)GT<F>::AW * (code & 0x7fffffffu));
        if (code & 0x80000000u) p.y = f_neg(p.y);
        acc = jac_add_mixed(acc, p, false);
    }
    jac_store<F>(buckets + (size_t)GT<F>::JW * ((size_t)lane * B + b), acc);
}

// G2 variant: one bucket per lane PAIR (fq2p.h).  Same algorithm, same memory formats.
template <class FP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate_pair(
    const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B, size_t sorted_stride, u64* buckets) {
    size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const size_t b = perm[(size_t)lane * B + t];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    Jac<FP> acc = Jac<FP>::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        Affine<FP> p = aff_load<FP>(pts + (size_t)24 * (code & 0x7fffffffu));
        if (code & 0x80000000u) p.y = f_neg(p.y);
        acc = jac_add_mixed(acc, p, false);
    }
    jac_store<FP>(buckets + (size_t)36 * ((size_t)lane * B + b), acc);
}

}  // namespace czk

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
                                                   size_t B, size_t sorted_stride, u64* buckets) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned lane = blockIdx.y;
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

}  // namespace czk

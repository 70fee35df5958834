// msm_acc.h -- the MSM's hot kernel: bucket accumulation, one thread per bucket.
// Included by msm_acc_g1.hip (F = Fq) and msm_acc_g2.hip (F = Fq2); those translation units are built with
// the Montgomery multiply inlined.  See msm.hip for the surrounding algorithm.
#pragma once
#include "czk_internal.h"

namespace czk {

// sorted[off .. off+cnt) lists this bucket's points as (w * n_bases + i) | sign<<31; pts holds the window
// multiples 2^(c*w) * P_i in affine Montgomery form.  acc += (+/-) P in XYZZ coordinates (curve.h; same edge cases as
// short_weierstrass_jacobian.rs:570-597).
template <class F>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate(const u64* pts, const u32* sorted, const u32* offsets, const u32* counts,
                                                   const u32* perm, size_t B, size_t sorted_stride, u64* buckets) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const size_t b = perm[(size_t)lane * B + t];   // buckets in descending-population order: equal work per wave
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    F ax = F::one(), ay = F::one(), azz = F::zero(), azzz = F::zero();   // XYZZ infinity
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        Affine<F> p = aff_load<F>(pts + (size_t)GT<F>::AW * (code & 0x7fffffffu));
        if (code & 0x80000000u) p.y = f_neg(p.y);
        xyzz_acc_mixed(ax, ay, azz, azzz, p.x, p.y);
    }
    xyzz_store<F>(buckets + (size_t)GT<F>::XW * ((size_t)lane * B + b), XYZZ<F>{ax, ay, azz, azzz});
}

// ------------------------------------------------------------------------------------------------
// bucket reduction: total = sum_j j * P_j + sum_j E_j over a segment; one level shrinks it by L.
//   P_out[m] = sum_{t in chunk m} P[t]
//   E_out[m] = sum_{t in chunk m} E[t] + 2^scale_dbl * sum_{t in chunk m} (t - start_m) * P[t]
// with 2^scale_dbl = L^level, so that  L^(level+1) * sum_m m P_out[m] + sum_m E_out[m]  is unchanged.
// ------------------------------------------------------------------------------------------------
template <class F, int JW, int SHIFT>   // JW = u64 words per XYZZ point; SHIFT = 1: one chunk per lane pair (F = Fq2P)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_reduce_level(const u64* P_in, const u64* E_in, size_t n_in, unsigned L, unsigned scale_dbl,
                                                     u64* P_out, u64* E_out, size_t n_out) {
    size_t m = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> SHIFT;
    if (m >= n_out) return;
    const size_t seg = blockIdx.y;
    const u64* P = P_in + (size_t)JW * seg * n_in;
    size_t start = m * L, end = start + L < n_in ? start + L : n_in;
    XYZZ<F> running = XYZZ<F>::zero(), A = XYZZ<F>::zero();
    for (size_t t = end; t-- > start;) {
        running = xyzz_add(running, xyzz_load<F>(P + JW * t));
        if (t > start) A = xyzz_add(A, running);
    }
    for (unsigned k = 0; k < scale_dbl; k++) A = xyzz_double(A);
    if (E_in) {
        const u64* E = E_in + (size_t)JW * seg * n_in;
        for (size_t t = start; t < end; t++) A = xyzz_add(A, xyzz_load<F>(E + JW * t));
    }
    xyzz_store<F>(P_out + (size_t)JW * (seg * n_out + m), running);
    xyzz_store<F>(E_out + (size_t)JW * (seg * n_out + m), A);
}

// out[seg] = P[seg] + E[seg]   (weights are b+1: sum_b (b+1) B_b = sum_b b B_b + sum_b B_b)
template <class F, int JW, int SHIFT>
__global__ void k_finish(const u64* P, const u64* E, size_t segs, u64* out) {
    size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> SHIFT;
    if (s >= segs) return;
    XYZZ<F> r = xyzz_load<F>(P + (size_t)JW * s);
    if (E) r = xyzz_add(r, xyzz_load<F>(E + (size_t)JW * s));
    jac_store<F>(out + (size_t)(JW / 4 * 3) * s, xyzz_to_jac(r));   // leave as the reference's Jacobian triple
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
    XYZZ<FP> acc = XYZZ<FP>::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        Affine<FP> p = aff_load<FP>(pts + (size_t)24 * (code & 0x7fffffffu));
        if (code & 0x80000000u) p.y = f_neg(p.y);
        acc = xyzz_add_mixed(acc, p);
    }
    xyzz_store<FP>(buckets + (size_t)48 * ((size_t)lane * B + b), acc);
}

}  // namespace czk

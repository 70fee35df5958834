// msm_acc.h -- the MSM's hot kernel: bucket accumulation, one thread per bucket.
// Included by msm_acc_g1.hip (F = Fq) and msm_acc_g2.hip (F = Fq2); those translation units are built with
// the Montgomery multiply inlined.  See msm.hip for the surrounding algorithm.
#pragma once
#ifdef CZK_FIX_NARROW   // A/B builds only: the fix-up kernels limited to 128 VGPRs (see k_accumulate_u_fix)
#define CZK_FIX_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define CZK_FIX_ATTR
#endif
#include "czk_internal.h"

namespace czk {

// blocks per lane of the fix-up kernels (see k_accumulate_u_fix)
constexpr unsigned FIX_GRID = 128;
static inline unsigned fix_grid(size_t B) {
    size_t g = (B + 127) / 128;
    return (unsigned)(g < FIX_GRID ? g : FIX_GRID);
}

// Which bucket a thread takes.  The buckets of a lane are walked in the order of `perm` (by size, so that the threads of a wave finish together); with
// G > 1 the lanes of an MSM are INTERLEAVED in groups of G: neighbouring threads take the same rank of neighbouring lanes (blockIdx.y selects the
// group, the last group may be smaller), so the G lanes' entry lists and table gathers of one rank are issued by the same wave.  Measured on the
// Groth16 step (4 lanes): the G1 kernel 10.4 -> 8.7 ms per launch, the G2 kernel 32.4 -> 29.5 ms, with every instruction and memory counter unchanged and the
// wave slots 93 % instead of 74 % occupied: what matters is that the launch walks ONE descending sequence of bucket sizes instead of one per lane
// (interleaving at workgroup granularity gains the same; the lane-after-lane order on a one-dimensional grid, an ascending or a scattered order do not:
// EXPERIMENTS.md section 14).  The rule that picks G is czk_internal.h acc_interleave.
__device__ __forceinline__ bool acc_work_item(size_t B, unsigned G, unsigned lanes, size_t& t, unsigned& lane) {
    const unsigned base = blockIdx.y * G, g = lanes - base < G ? lanes - base : G;
    const size_t lin = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    t = g == 1 ? lin : lin / g;
    lane = base + (g == 1 ? 0u : (unsigned)(lin % g));
    return t < B;
}

// sorted[off .. off+cnt) lists this bucket's points as (w * n_bases + i) | sign<<31; pts holds the window
// multiples 2^(c*w) * P_i in affine Montgomery form.  acc += (+/-) P in XYZZ coordinates (curve.h; same edge cases as
// short_weierstrass_jacobian.rs:570-597).
// ------------------------------------------------------------------------------------------------
// Over-full buckets.  One thread folds one bucket, so a bucket with a million points would be a million dependent
// additions (seconds) -- and real witnesses do that: boolean wires put half of the scalars on digit 1 of window 0 (the
// reference special-cases scalar == 1 for the same reason, variable_base.rs:44-48).  The main kernels therefore stop after
// HEAVY_CHUNK entries; every further chunk of HEAVY_SUB entries becomes a work item folded by its own thread into a
// partial sum (saturated formulas: complete, no exception list), and k_heavy_combine tree-adds a bucket's partials to it.
// With the benchmark's uniformly random scalars no bucket comes near the limit and the item list stays empty.
// ------------------------------------------------------------------------------------------------
constexpr u32 HEAVY_CHUNK = 1024, HEAVY_SUB = 256;   // main kernels fold the first 1024 entries; the rest in 256-entry work items
// items: (lane, bucket, chunk index j); heavy: (lane, bucket, first item, number of items); hdr[0] = #items, hdr[1] = #heavy buckets
template <class F>   // (template only so that each translation unit gets its own instance)
__global__ void k_heavy_list(const u32* counts, size_t B, u32* hdr, u32* items, u32* heavy, u32 cap) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned lane = blockIdx.y;
    u32 cnt = counts[(size_t)lane * B + b];
    if (cnt <= HEAVY_CHUNK) return;
    u32 extra = (cnt - HEAVY_CHUNK + HEAVY_SUB - 1) / HEAVY_SUB;
    u32 base = atomicAdd(&hdr[0], extra);
    u32 slot = atomicAdd(&hdr[1], 1u);
    if (base + extra > cap || slot >= cap) {             // cannot happen: cap >= total entries / HEAVY_SUB + 64; flagged anyway
        atomicOr(&hdr[2], 1u);
        return;
    }
    heavy[4 * slot] = lane;
    heavy[4 * slot + 1] = (u32)b;
    heavy[4 * slot + 2] = base;
    heavy[4 * slot + 3] = extra;
    for (u32 j = 0; j < extra; j++) {
        items[3 * (base + j)] = lane;
        items[3 * (base + j) + 1] = (u32)b;
        items[3 * (base + j) + 2] = j;
    }
}
template <class F>
__device__ __forceinline__ F heavy_coord(const u64* p, bool unsat);   // table coordinate -> saturated Montgomery form
template <>
__device__ __forceinline__ Fq heavy_coord<Fq>(const u64* p, bool unsat) {
    Fq v = fp_load<FqParams>(p);
    return unsat ? fp_mul(v, fqu_k_from_u()) : v;
}
template <>
__device__ __forceinline__ Fq2 heavy_coord<Fq2>(const u64* p, bool unsat) {
    return Fq2{heavy_coord<Fq>(p, unsat), heavy_coord<Fq>(p + 6, unsat)};
}
// A bucket slot <-> the saturated XYZZ form, for the rare-path kernels (over-full buckets, dirty buckets, deferred points): when the
// reduction runs in the unsaturated residue system (G1, `ubuckets`), slots hold u-form coordinates (fqu.h) and are converted here.
template <class F>
__device__ __forceinline__ XYZZ<F> bucket_load_sat(const u64* slot, int ubuckets) {
    if (!ubuckets) return xyzz_load<F>(slot);
    constexpr int FW = GT<F>::AW / 2;
    XYZZ<F> r{heavy_coord<F>(slot, true), heavy_coord<F>(slot + FW, true), heavy_coord<F>(slot + 2 * FW, true), heavy_coord<F>(slot + 3 * FW, true)};
    return r;   // (a zero zz stays zero: infinity survives the conversion)
}
__device__ __forceinline__ Fq coord_to_u(const Fq& v) { return fp_mul(v, fqu_k_to_u()); }
__device__ __forceinline__ Fq2 coord_to_u(const Fq2& v) { return Fq2{coord_to_u(v.c0), coord_to_u(v.c1)}; }
template <class F>
__device__ __forceinline__ void bucket_store_sat(u64* slot, const XYZZ<F>& v, int ubuckets) {
    if (!ubuckets) {
        xyzz_store<F>(slot, v);
        return;
    }
    XYZZ<F> u = v.is_zero() ? XYZZ<F>{F::zero(), F::zero(), F::zero(), F::zero()} : XYZZ<F>{coord_to_u(v.x), coord_to_u(v.y), coord_to_u(v.zz), coord_to_u(v.zzz)};
    xyzz_store<F>(slot, u);
}

template <class F>
// (<= 128 VGPRs, spilling: with no work items -- the normal case -- its blocks must slip into the register space the accumulate
// kernels leave free; at 250 VGPRs each empty block waited for a drained SIMD and the empty launch took 7 ms in the pipeline)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_accumulate_heavy(const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B,
                                                         size_t sorted_stride, const u32* hdr, const u32* items, u64* partials, u32 cap, int unsat) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    u32 n_items = hdr[0] < cap ? hdr[0] : cap;
    if (k >= n_items) return;
    const u32 lane = items[3 * k], b = items[3 * k + 1], j = items[3 * k + 2];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    const u32 first = HEAVY_CHUNK + j * HEAVY_SUB;
    u32 off = offsets[(size_t)lane * B + b] + first, cnt = counts[(size_t)lane * B + b] - first;
    if (cnt > HEAVY_SUB) cnt = HEAVY_SUB;
    constexpr int FW = GT<F>::AW / 2;
    F ax = F::one(), ay = F::one(), azz = F::zero(), azzz = F::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        const u64* pp = pts + (size_t)GT<F>::AW * (code & 0x7fffffffu);
        F qx = heavy_coord<F>(pp, unsat != 0), qy = heavy_coord<F>(pp + FW, unsat != 0);
        if (code & 0x80000000u) qy = f_neg(qy);
        xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy);
    }
    xyzz_store<F>(partials + (size_t)GT<F>::XW * k, XYZZ<F>{ax, ay, azz, azzz});
}
// one 128-thread block per over-full bucket: strided sums of its partials, tree reduction (in place in `partials`), add to the bucket
template <class F>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_heavy_combine(const u32* hdr, const u32* heavy, u64* partials, size_t B, u64* buckets, const uint8_t* dirty, u32 cap, int ubuckets) {
    const u32 k = blockIdx.x, tid = threadIdx.x;
    u32 n_heavy = hdr[1] < cap ? hdr[1] : cap;
    if (k >= n_heavy) return;
    const u32 lane = heavy[4 * k], b = heavy[4 * k + 1], base = heavy[4 * k + 2], extra = heavy[4 * k + 3];
    if (dirty && dirty[(size_t)lane * B + b]) return;   // k_accumulate_u*_fix recomputes a dirty bucket from ALL of its entries
    constexpr int XW = GT<F>::XW;
    u64* part = partials + (size_t)XW * base;
    XYZZ<F> v = XYZZ<F>::zero();
    for (u32 j = tid; j < extra; j += 128) v = xyzz_add(v, xyzz_load<F>(part + (size_t)XW * j));
    __syncthreads();                                     // every partial has been read before slots 0..127 are reused
    if (tid < extra) xyzz_store<F>(part + (size_t)XW * tid, v);
    __syncthreads();
    const u32 live = extra < 128 ? extra : 128;
    for (u32 stride = 64; stride > 0; stride >>= 1) {
        if (tid < stride && tid + stride < live) {
            v = xyzz_add(v, xyzz_load<F>(part + (size_t)XW * (tid + stride)));
            xyzz_store<F>(part + (size_t)XW * tid, v);
        }
        __syncthreads();
    }
    if (tid == 0) {
        u64* slot = buckets + (size_t)XW * ((size_t)lane * B + b);
        bucket_store_sat<F>(slot, xyzz_add(bucket_load_sat<F>(slot, ubuckets), v), ubuckets);
    }
}

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
    if (cnt > HEAVY_CHUNK) cnt = HEAVY_CHUNK;   // the rest of an over-full bucket is folded by k_accumulate_heavy
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


// ------------------------------------------------------------------------------------------------
// Tail of the bucket reduction, used once a level has n_in <= 1024 entries per lane.  The chunked levels below that
// point are latency: a handful of threads each walking ~30 dependent point additions per level.  Here
//   sum_j j P[j] = sum_b 2^b S_b,   S_b = sum_{j : bit b of j set} P[j]
// so blocks 0..9 each form one S_b, block 10 sums E and block 11 sums P, all as depth-10 tree reductions in parallel;
// k_reduce_tail_finish then runs the 10-step Horner recombination.  total = 2^scale * sum_j j P[j] + sum E + sum P.
// ------------------------------------------------------------------------------------------------
constexpr unsigned TAIL_MAX = 1024, TAIL_BLOCKS = 12, TAIL_THREADS = 128;   // 128-thread blocks: a 512-thread block needs a whole idle CU at ~250 VGPRs
template <class F, int JW, int SHIFT>   // SHIFT = 1: one point per lane pair (F = Fq2P), as in k_reduce_level
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_reduce_tail_sums(const u64* P_in, const u64* E_in, size_t n_in,
                                                                                                      u64* scratch, u64* sums) {
    const unsigned which = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x >> SHIFT;
    constexpr unsigned NT = TAIL_THREADS >> SHIFT;   // points in flight per block
    const u64* src = which == 10 ? E_in : P_in;
    XYZZ<F> v = XYZZ<F>::zero();
    if (src) {
        src += (size_t)JW * seg * n_in;
        for (size_t j = tid; j < n_in; j += NT)
            if (which >= 10 || ((j >> which) & 1)) v = xyzz_add(v, xyzz_load<F>(src + JW * j));
    }
    u64* sc = scratch + (size_t)JW * ((size_t)(seg * TAIL_BLOCKS + which) * TAIL_THREADS);
    xyzz_store<F>(sc + (size_t)JW * tid, v);
    __syncthreads();
    for (unsigned stride = NT / 2; stride > 0; stride >>= 1) {
        if (tid < stride) {
            v = xyzz_add(v, xyzz_load<F>(sc + (size_t)JW * (tid + stride)));
            xyzz_store<F>(sc + (size_t)JW * tid, v);
        }
        __syncthreads();
    }
    if (tid == 0) xyzz_store<F>(sums + (size_t)JW * (seg * TAIL_BLOCKS + which), v);
}
template <class F, int JW, int SHIFT>
__global__ void k_reduce_tail_finish(const u64* sums, unsigned scale_dbl, size_t segs, u64* out) {
    size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> SHIFT;
    if (s >= segs) return;
    const u64* sm = sums + (size_t)JW * s * TAIL_BLOCKS;
    XYZZ<F> acc = xyzz_load<F>(sm + JW * 9);
    for (int b = 8; b >= 0; b--) acc = xyzz_add(xyzz_double(acc), xyzz_load<F>(sm + JW * b));
    for (unsigned k = 0; k < scale_dbl; k++) acc = xyzz_double(acc);
    acc = xyzz_add(acc, xyzz_load<F>(sm + JW * 10));
    acc = xyzz_add(acc, xyzz_load<F>(sm + JW * 11));   // weights are b + 1 (see k_finish)
    jac_store<F>(out + (size_t)(JW / 4 * 3) * s, xyzz_to_jac(acc));
}

// ------------------------------------------------------------------------------------------------
// The bucket reduction in the unsaturated residue system: same algorithm as k_reduce_level / k_reduce_tail_* above, with the buckets and
// every intermediate array in u-form.  PT = the point type's operations: XyzzOps (fqu.h XYZZU, short Weierstrass G1), TeOps (te.h TEU,
// twisted Edwards G1: unified additions, no infinity flag) or Xyzz2Ops (fq2pu.h XYZZU2, G2: one point per lane PAIR, PT::SHIFT = 1).
// PT::JW = u64 words per point, PT::JACW = u64 words of the Jacobian result.
// ------------------------------------------------------------------------------------------------
template <class PT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(PT::WAVES, PT::WAVES))) void k_reduce_level_p(const u64* P_in, const u64* E_in, size_t n_in, unsigned L,
                                                                                                    unsigned scale_dbl, u64* P_out, u64* E_out, size_t n_out) {
    constexpr int JW = PT::JW;
    size_t m = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> PT::SHIFT;
    if (m >= n_out) return;
    const size_t seg = blockIdx.y;
    const u64* P = P_in + (size_t)JW * seg * n_in;
    size_t start = m * L, end = start + L < n_in ? start + L : n_in;
    typename PT::P running = PT::zero(), A = PT::zero();
    for (size_t t = end; t-- > start;) {
        PT::add(running, PT::load(P + JW * t));
        if (t > start) PT::add(A, running);
    }
    for (unsigned k = 0; k < scale_dbl; k++) PT::dbl(A);
    if (E_in) {
        const u64* E = E_in + (size_t)JW * seg * n_in;
        for (size_t t = start; t < end; t++) PT::add(A, PT::load(E + JW * t));
    }
    PT::store(P_out + (size_t)JW * (seg * n_out + m), running);
    PT::store(E_out + (size_t)JW * (seg * n_out + m), A);
}
template <class PT>
__global__ void k_finish_p(const u64* P, const u64* E, size_t segs, u64* out) {
    constexpr int JW = PT::JW;
    size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> PT::SHIFT;
    if (s >= segs) return;
    typename PT::P r = PT::load(P + (size_t)JW * s);
    if (E) PT::add(r, PT::load(E + (size_t)JW * s));
    PT::store_jac(out + (size_t)PT::JACW * s, r);   // the result leaves as the reference's Jacobian triple, Montgomery form
}
template <class PT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(PT::WAVES, PT::WAVES))) void k_reduce_tail_sums_p(const u64* P_in, const u64* E_in, size_t n_in, u64* scratch,
                                                                                                        u64* sums) {
    constexpr int JW = PT::JW;
    constexpr unsigned NT = TAIL_THREADS >> PT::SHIFT;   // points in flight per block
    const unsigned which = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x >> PT::SHIFT;
    const u64* src = which == 10 ? E_in : P_in;
    typename PT::P v = PT::zero();
    if (src) {
        src += (size_t)JW * seg * n_in;
        for (size_t j = tid; j < n_in; j += NT)
            if (which >= 10 || ((j >> which) & 1)) PT::add(v, PT::load(src + JW * j));
    }
    u64* sc = scratch + (size_t)JW * ((size_t)(seg * TAIL_BLOCKS + which) * TAIL_THREADS);
    PT::store(sc + (size_t)JW * tid, v);
    __syncthreads();
    for (unsigned stride = NT / 2; stride > 0; stride >>= 1) {
        if (tid < stride) {
            PT::add(v, PT::load(sc + (size_t)JW * (tid + stride)));
            PT::store(sc + (size_t)JW * tid, v);
        }
        __syncthreads();
    }
    if (tid == 0) PT::store(sums + (size_t)JW * (seg * TAIL_BLOCKS + which), v);
}
template <class PT>
__global__ void k_reduce_tail_finish_p(const u64* sums, unsigned scale_dbl, size_t segs, u64* out) {
    constexpr int JW = PT::JW;
    size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> PT::SHIFT;
    if (s >= segs) return;
    const u64* sm = sums + (size_t)JW * s * TAIL_BLOCKS;
    typename PT::P acc = PT::load(sm + JW * 9);
    for (int b = 8; b >= 0; b--) {
        PT::dbl(acc);
        PT::add(acc, PT::load(sm + JW * b));
    }
    for (unsigned k = 0; k < scale_dbl; k++) PT::dbl(acc);
    PT::add(acc, PT::load(sm + JW * 10));
    PT::add(acc, PT::load(sm + JW * 11));   // weights are b + 1 (see k_finish)
    PT::store_jac(out + (size_t)PT::JACW * s, acc);
}

#ifdef CZK_FQU_G1
// G1 accumulation in the unsaturated residue system (fqu.h): `pts` holds x R' mod p, y R' mod p as canonical
// 12 x u32 integers (converted at registration).  Buckets leave in the usual saturated XYZZ Montgomery form.
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate_u(
    const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B, size_t sorted_stride, u64* buckets,
    uint8_t* dirty, u32* exc_count, u32* exc_list, u32 exc_cap, int ubuckets) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const size_t b = perm[(size_t)lane * B + t];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    FqU ax, ay, azz, azzz;
    bool inf = true;
    if (cnt > HEAVY_CHUNK) cnt = HEAVY_CHUNK;   // the rest of an over-full bucket is folded by k_accumulate_heavy
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        const u64* pp = pts + (size_t)12 * (code & 0x7fffffffu);
        FqU qx = fqu_unpack(fp_load<FqParams>(pp));
        FqU qy = fqu_unpack(fp_load<FqParams>(pp + 6));
        if (code & 0x80000000u) {
#pragma unroll
            for (int i = 0; i < 14; i++) qy.l[i] = fqu_4p(i) - qy.l[i];   // 4p - y, lazy limbs < 2^29
        }
        if (inf) {
            ax = qx;
            ay = fqu_normalize(qy);
            azz = fqu_one();
            azzz = azz;
            inf = false;
            continue;
        }
        if (!fqu_xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy)) {
            // possible P == +-Q (about 1 in 2^24 additions for honest inputs).  Addition commutes, so the point is
            // deferred: k_accumulate_u_cleanup adds it to the finished bucket with the saturated formulas (all edge
            // cases).  If the list is full (adversarial inputs), the whole bucket goes to k_accumulate_u_fix instead.
            u32 slot = atomicAdd(exc_count, 1u);
            if (slot < exc_cap) {
                exc_list[3 * slot] = lane;
                exc_list[3 * slot + 1] = (u32)b;
                exc_list[3 * slot + 2] = code;
                continue;
            }
            dirty[(size_t)lane * B + b] = 1;
            return;
        }
    }
    u64* slot = buckets + (size_t)24 * ((size_t)lane * B + b);
    if (ubuckets) {   // the reduction runs in this residue system too (k_reduce_*_u): no conversion, the accumulator is stored as it is
        XYZZU o;
        o.inf = inf;
        o.x = ax;
        o.y = ay;
        o.zz = azz;
        o.zzz = azzz;
        xyzzu_store(slot, o);
        return;
    }
    XYZZ<Fq> out = XYZZ<Fq>::zero();
    if (!inf) {
        const Fq kf = fqu_k_from_u();
        out.x = fp_mul(fqu_pack(ax), kf);
        out.y = fp_mul(fqu_pack(ay), kf);
        out.zz = fp_mul(fqu_pack(azz), kf);
        out.zzz = fp_mul(fqu_pack(azzz), kf);
    }
    xyzz_store<Fq>(slot, out);
}

// recomputes the buckets k_accumulate_u gave up on, in the saturated residue system (points converted on the fly).
// (Normally no bucket is dirty and every block returns at once; at ~250 VGPRs each block still has to wait for a drained SIMD
// next to the accumulate kernels -- 2.3 ms on average per empty launch in the pipeline, profiles/r02_kernel_trace_stats.txt.
// Limiting it to 128 VGPRs (-DCZK_FIX_NARROW) removes that wait and shortens one proof's latency by ~1 ms, but the blocks then
// run BESIDE the accumulate waves and cost 0.8 % of throughput (A/B on one box: 89.8 vs 89.1 ms per proof): not adopted.
// What is adopted: a grid of at most FIX_GRID blocks per lane striding over the buckets, so an all-clean launch places a few
// hundred wide waves instead of tens of thousands.)
__global__ __launch_bounds__(128) CZK_FIX_ATTR void k_accumulate_u_fix(const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B,
                                                         size_t sorted_stride, u64* buckets, const uint8_t* dirty, int ubuckets) {
    const unsigned lane = blockIdx.y;
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    const Fq kf = fqu_k_from_u();
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (size_t)gridDim.x * blockDim.x) {
    if (!dirty[(size_t)lane * B + b]) continue;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    Fq ax = Fq::one(), ay = Fq::one(), azz = Fq::zero(), azzz = Fq::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        const u64* pp = pts + (size_t)12 * (code & 0x7fffffffu);
        Fq qx = fp_mul(fp_load<FqParams>(pp), kf), qy = fp_mul(fp_load<FqParams>(pp + 6), kf);
        if (code & 0x80000000u) qy = fp_neg(qy);
        xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy);
    }
    bucket_store_sat<Fq>(buckets + (size_t)24 * ((size_t)lane * B + b), XYZZ<Fq>{ax, ay, azz, azzz}, ubuckets);
    }
}

// adds the deferred points into the finished buckets (sequentially: several may hit one bucket); buckets that were
// recomputed from scratch by k_accumulate_u_fix already contain theirs
__global__ void k_accumulate_u_cleanup(const u64* pts, size_t B, u64* buckets, const uint8_t* dirty, const u32* exc_count, const u32* exc_list,
                                       u32 exc_cap, int ubuckets) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    u32 n = *exc_count;
    if (n > exc_cap) n = exc_cap;
    const Fq kf = fqu_k_from_u();
    for (u32 k = 0; k < n; k++) {
        u32 lane = exc_list[3 * k], b = exc_list[3 * k + 1], code = exc_list[3 * k + 2];
        if (dirty[(size_t)lane * B + b]) continue;
        u64* slot = buckets + (size_t)24 * ((size_t)lane * B + b);
        XYZZ<Fq> acc = bucket_load_sat<Fq>(slot, ubuckets);
        const u64* pp = pts + (size_t)12 * (code & 0x7fffffffu);
        Fq qx = fp_mul(fp_load<FqParams>(pp), kf), qy = fp_mul(fp_load<FqParams>(pp + 6), kf);
        if (code & 0x80000000u) qy = fp_neg(qy);
        xyzz_acc_mixed(acc.x, acc.y, acc.zz, acc.zzz, qx, qy);
        bucket_store_sat<Fq>(slot, acc, ubuckets);
    }
}

// one-time conversion of registered window tables (G1 and G2): Fq coordinate <- coordinate * R' / R  (canonical integer)
__global__ void k_convert_to_u(u64* pts, size_t n_coords) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coords) return;
    fp_store<FqParams>(pts + 6 * i, fp_mul(fp_load<FqParams>(pts + 6 * i), fqu_k_to_u()));
}
__global__ void k_convert_from_u(u64* pts, size_t n_coords) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_coords) return;
    fp_store<FqParams>(pts + 6 * i, fp_mul(fp_load<FqParams>(pts + 6 * i), fqu_k_from_u()));
}

// ---- the G1 point types of the u-form bucket reduction (k_reduce_*_p above) ----
struct XyzzOps {
    static constexpr int WAVES = 2, JW = 24, JACW = 18, SHIFT = 0;
    typedef XYZZU P;
    static __device__ __forceinline__ P zero() { return xyzzu_zero(); }
    static __device__ __forceinline__ P load(const u64* p) { return xyzzu_load(p); }
    static __device__ __forceinline__ void store(u64* p, const P& a) { xyzzu_store(p, a); }
    static __device__ __forceinline__ void add(P& a, const P& b) { xyzzu_add(a, b); }
    static __device__ __forceinline__ void dbl(P& a) { xyzzu_double(a); }
    static __device__ __forceinline__ void store_jac(u64* out, const P& a) { jac_store<Fq>(out, xyzz_to_jac(xyzzu_to_sat(a))); }
};
#ifndef CZK_TERED_WAVES
#define CZK_TERED_WAVES 3   // (A/B builds)
#endif
struct TeOps {
    // 3 waves per SIMD = at most 168 VGPRs: a reduction wave then fits NEXT to the two resident waves of k_accumulate_te (155 VGPRs
    // each) instead of taking one of their slots (same-box A/B: 80.6 against 81.2 ms per proof)
    static constexpr int WAVES = CZK_TERED_WAVES, JW = 24, JACW = 18, SHIFT = 0;
    typedef TEU P;
    static __device__ __forceinline__ P zero() { return teu_identity(); }
    static __device__ __forceinline__ P load(const u64* p) { return teu_load(p); }
    static __device__ __forceinline__ void store(u64* p, const P& a) { teu_store(p, a); }
    static __device__ __forceinline__ void add(P& a, const P& b) { teu_add(a, b); }
    static __device__ __forceinline__ void dbl(P& a) { teu_double(a); }
    static __device__ __forceinline__ void store_jac(u64* out, const P& a) { jac_store<Fq>(out, teu_to_jac(a)); }
};
// ---- twisted Edwards kernels (te.h): table conversion, bucket accumulation, over-full buckets -------------------------------------
// SW affine Montgomery points (12 u64 + infinity flag) -> (Y - X, Y + X, 2 D X Y) as 3 x 16 u32 limb arrays (te.h: 24 u64 per point).  With w = (x + 1) / s:
//   X = f w / y,  Y = (w - 1) / (w + 1)   ->   one shared denominator y (w + 1) per point, Montgomery's trick over CH points per thread.
// A point without an image (y = 0 or w = -1: even order, never in G1) raises *bad; infinity flags stay with the caller (such entries
// are never referenced: their digits are dropped).  scratch: n Fq.
__global__ __launch_bounds__(128) void k_sw_to_te_niels(const u64* aff, const uint8_t* inf, size_t n, unsigned CH, u64* scratch, u64* out, u32* bad) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = t * CH;
    if (start >= n) return;
    size_t end = start + CH < n ? start + CH : n;
    constexpr u32 sim[12] = TE_S_INV_S, fm[12] = TE_F_S, dm[12] = TE_2D_S;
    Fq s_inv, f, d2;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        s_inv.l[i] = sim[i];
        f.l[i] = fm[i];
        d2.l[i] = dm[i];
    }
    const Fq one = Fq::one();
    Fq acc = one;
    for (size_t i = start; i < end; i++) {
        fp_store<FqParams>(scratch + 6 * i, acc);
        if (inf[i]) continue;
        const Fq w = fp_mul(fp_add(fp_load<FqParams>(aff + 12 * i), one), s_inv);
        const Fq den = fp_mul(fp_load<FqParams>(aff + 12 * i + 6), fp_add(w, one));
        if (den.is_zero()) {
            atomicOr(bad, 1u);
            continue;
        }
        acc = fp_mul(acc, den);
    }
    Fq inv = fp_inv(acc);
    const Fq kt = fqu_k_to_u();
    for (size_t i = end; i-- > start;) {
        u64* o = out + (size_t)TE_POINT_U64 * i;
        Fq ym = one, yp = one, k2 = Fq::zero();                       // the neutral element (0, 1) for entries that are never used
        if (!inf[i]) {
            const Fq y = fp_load<FqParams>(aff + 12 * i + 6);
            const Fq w = fp_mul(fp_add(fp_load<FqParams>(aff + 12 * i), one), s_inv);
            const Fq wp = fp_add(w, one), den = fp_mul(y, wp);
            if (!den.is_zero()) {
                const Fq dinv = fp_mul(inv, fp_load<FqParams>(scratch + 6 * i));
                inv = fp_mul(inv, den);
                const Fq X = fp_mul(fp_mul(fp_mul(f, w), wp), dinv);      // f w (w + 1) / (y (w + 1))
                const Fq Y = fp_mul(fp_mul(fp_sub(w, one), y), dinv);     // (w - 1) y / (y (w + 1))
                ym = fp_sub(Y, X);
                yp = fp_add(Y, X);
                k2 = fp_mul(d2, fp_mul(X, Y));
            }
        }
        te_store_coord(o, fqu_unpack(fp_mul(ym, kt)));
        te_store_coord(o + 8, fqu_unpack(fp_mul(yp, kt)));
        te_store_coord(o + 16, fqu_unpack(fp_mul(k2, kt)));
    }
}

// bucket accumulation, one thread per bucket: unified additions, so no exception list, no dirty flags, no infinity flag; the first
// entry is taken over with one multiplication (teu_from_niels)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate_te(const u64* pts, const u32* sorted, const u32* offsets,
                                                                                                   const u32* counts, const u32* perm, size_t B, size_t sorted_stride,
                                                                                                   u64* buckets, unsigned G, unsigned lanes) {
    size_t t;
    unsigned lane;
    if (!acc_work_item(B, G, lanes, t, lane)) return;
    const size_t b = perm[(size_t)lane * B + t];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    if (cnt > HEAVY_CHUNK) cnt = HEAVY_CHUNK;   // the rest of an over-full bucket is folded by k_accumulate_heavy_te
    TEU acc = teu_identity();
    for (u32 e = 0; e < cnt; e++) {
        const u32 code = srt[off + e];
        FqU ym, yp, k2;
#ifdef CZK_TE_IDX_MASK   // TIMING EXPERIMENT ONLY (wrong results): confine the gathers to a cache-resident part of the table
        te_load_niels(pts + (size_t)TE_POINT_U64 * (code & CZK_TE_IDX_MASK), (code & 0x80000000u) != 0, ym, yp, k2);
#else
        te_load_niels(pts + (size_t)TE_POINT_U64 * (code & 0x7fffffffu), (code & 0x80000000u) != 0, ym, yp, k2);
#endif
        if (e == 0) acc = teu_from_niels(ym, yp, k2);
#ifdef CZK_TE_IL   // A/B builds: the products of an addition as 3 - 4 interleaved multiply-add chains (fqu_il.h: measured no faster, 206 VGPRs)
        else teu_madd_il(acc, ym, yp, k2);
#else
        else teu_madd(acc, ym, yp, k2);
#endif
    }
    teu_store(buckets + (size_t)24 * ((size_t)lane * B + b), acc);
}
// over-full buckets (see k_accumulate_heavy / k_heavy_combine above): the same work items, folded and combined with the unified law
// (<= 128 VGPRs like k_accumulate_heavy: the normally empty launches must not wait for a drained SIMD beside the accumulate kernels)
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_accumulate_heavy_te(const u64* pts, const u32* sorted, const u32* offsets,
                                                                                                         const u32* counts, size_t B, size_t sorted_stride, const u32* hdr,
                                                                                                         const u32* items, u64* partials, u32 cap) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    u32 n_items = hdr[0] < cap ? hdr[0] : cap;
    if (k >= n_items) return;
    const u32 lane = items[3 * k], b = items[3 * k + 1], j = items[3 * k + 2];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    const u32 first = HEAVY_CHUNK + j * HEAVY_SUB;
    u32 off = offsets[(size_t)lane * B + b] + first, cnt = counts[(size_t)lane * B + b] - first;
    if (cnt > HEAVY_SUB) cnt = HEAVY_SUB;
    TEU acc = teu_identity();
    for (u32 e = 0; e < cnt; e++) {
        const u32 code = srt[off + e];
        FqU ym, yp, k2;
        te_load_niels(pts + (size_t)TE_POINT_U64 * (code & 0x7fffffffu), (code & 0x80000000u) != 0, ym, yp, k2);
        teu_madd(acc, ym, yp, k2);
    }
    teu_store(partials + (size_t)24 * k, acc);
}
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_heavy_combine_te(const u32* hdr, const u32* heavy, u64* partials, size_t B,
                                                                                                      u64* buckets, u32 cap) {
    const u32 k = blockIdx.x, tid = threadIdx.x;
    u32 n_heavy = hdr[1] < cap ? hdr[1] : cap;
    if (k >= n_heavy) return;
    const u32 lane = heavy[4 * k], b = heavy[4 * k + 1], base = heavy[4 * k + 2], extra = heavy[4 * k + 3];
    u64* part = partials + (size_t)24 * base;
    TEU v = teu_identity();
    for (u32 j = tid; j < extra; j += 128) teu_add(v, teu_load(part + (size_t)24 * j));
    __syncthreads();                                     // every partial has been read before slots 0..127 are reused
    if (tid < extra) teu_store(part + (size_t)24 * tid, v);
    __syncthreads();
    const u32 live = extra < 128 ? extra : 128;
    for (u32 stride = 64; stride > 0; stride >>= 1) {
        if (tid < stride && tid + stride < live) {
            teu_add(v, teu_load(part + (size_t)24 * (tid + stride)));
            teu_store(part + (size_t)24 * tid, v);
        }
        __syncthreads();
    }
    if (tid == 0) {
        u64* slot = buckets + (size_t)24 * ((size_t)lane * B + b);
        TEU acc = teu_load(slot);
        teu_add(acc, v);
        teu_store(slot, acc);
    }
}
#endif
#ifdef CZK_FQU_G2
// ---- G2: the same three kernels over Fq2U / Fq2 ---------------------------------------------------------------
__device__ __forceinline__ Fq2U fq2u_load(const u64* p) {
    return Fq2U{fqu_unpack(fp_load<FqParams>(p)), fqu_unpack(fp_load<FqParams>(p + 6))};
}
__device__ __forceinline__ void fq2u_store(u64* p, const Fq2U& a) {
    fp_store<FqParams>(p, fqu_pack(a.c0));
    fp_store<FqParams>(p + 6, fqu_pack(a.c1));
}
__device__ __forceinline__ Fq2 fq2u_to_sat(const Fq2U& a) {
    const Fq kf = fqu_k_from_u();
    return Fq2{fp_mul(fqu_pack(a.c0), kf), fp_mul(fqu_pack(a.c1), kf)};
}
__device__ __forceinline__ Fq2 fq2_from_table_u(const u64* p) {   // table coordinate (x R' mod p) -> saturated Montgomery
    const Fq kf = fqu_k_from_u();
    return Fq2{fp_mul(fp_load<FqParams>(p), kf), fp_mul(fp_load<FqParams>(p + 6), kf)};
}

#ifndef CZK_G2ACC_WAVES
#define CZK_G2ACC_WAVES 1   // (A/B builds)
#endif
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(CZK_G2ACC_WAVES, CZK_G2ACC_WAVES))) void k_accumulate_u2(
    const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B, size_t sorted_stride, u64* buckets,
    uint8_t* dirty, u32* exc_count, u32* exc_list, u32 exc_cap, int ubuckets, unsigned G, unsigned lanes) {
    size_t t;
    unsigned lane;
    if (!acc_work_item(B, G, lanes, t, lane)) return;
    const size_t b = perm[(size_t)lane * B + t];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    Fq2U ax, ay, azz, azzz;
    bool inf = true;
    if (cnt > HEAVY_CHUNK) cnt = HEAVY_CHUNK;   // the rest of an over-full bucket is folded by k_accumulate_heavy
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        const u64* pp = pts + (size_t)24 * (code & 0x7fffffffu);
        Fq2U qx = fq2u_load(pp), qy = fq2u_load(pp + 12);
        if (code & 0x80000000u) {
#pragma unroll
            for (int i = 0; i < 14; i++) {
                qy.c0.l[i] = fqu_4p(i) - qy.c0.l[i];
                qy.c1.l[i] = fqu_4p(i) - qy.c1.l[i];
            }
            if (inf) {   // (as an accumulator's Y it must be normalised; as a multiplicand the lazy form will do: limbs < 2^30)
                qy.c0 = fqu_normalize(qy.c0);
                qy.c1 = fqu_normalize(qy.c1);
            }
        }
        if (inf) {
            ax = qx;
            ay = qy;
            azz = Fq2U{fqu_one(), FqU{}};
#pragma unroll
            for (int i = 0; i < 14; i++) azz.c1.l[i] = 0;
            azzz = azz;
            inf = false;
            continue;
        }
#ifdef CZK_G2_IL   // A/B builds: products grouped into four interleaved multiply-add chains (fqu_il.h: measured no faster, 369 registers)
        if (!fq2u_xyzz_acc_mixed_il(ax, ay, azz, azzz, qx, qy)) {
#else
        if (!fq2u_xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy)) {
#endif
            u32 slot = atomicAdd(exc_count, 1u);
            if (slot < exc_cap) {
                exc_list[3 * slot] = lane;
                exc_list[3 * slot + 1] = (u32)b;
                exc_list[3 * slot + 2] = code;
                continue;
            }
            dirty[(size_t)lane * B + b] = 1;
            return;
        }
    }
    u64* slot = buckets + (size_t)48 * ((size_t)lane * B + b);
    if (ubuckets && !inf) {   // u-form for the reduction of fq2pu.h: the coordinates as they are (x < 85 p, y < 36 p, zz, zzz < 3 p: below 2^384)
        fq2u_store(slot, ax);
        fq2u_store(slot + 12, ay);
        fq2u_store(slot + 24, azz);
        fq2u_store(slot + 36, azzz);
        return;
    }
    XYZZ<Fq2> out = XYZZ<Fq2>::zero();
    if (!inf) {
        out.x = fq2u_to_sat(ax);
        out.y = fq2u_to_sat(ay);
        out.zz = fq2u_to_sat(azz);
        out.zzz = fq2u_to_sat(azzz);
    }
    xyzz_store<Fq2>(slot, out);
}

#ifdef CZK_FQ2PU
// The same accumulation with one bucket per lane PAIR (fq2pu.h): even lanes hold the c0 halves, odd lanes the c1 halves; two waves per SIMD.
// Buckets leave in u-form (the reduction of fq2pu.h); exceptional additions go to the same deferred list / dirty flags as above.
// BLOCK / WAVES: <128, 2> lets the kernel take what two waves per SIMD allow (204 registers); <512, 3> caps it at 168 registers and is
// launched with 82 KiB of (unused) dynamic LDS, so that exactly ONE workgroup -- two waves per SIMD, 336 registers -- fits a CU and a wave of
// the NTT / sort / reduction kernels (<= 168 registers, <= 78 KiB of LDS) still fits beside it.
template <int BLOCK, int WAVES>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_accumulate_u2p(
    const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B, size_t sorted_stride, u64* buckets,
    uint8_t* dirty, u32* exc_count, u32* exc_list, u32 exc_cap) {
    size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const bool par = pair_parity();
    const size_t b = perm[(size_t)lane * B + t];
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    FqU ax, ay, azz, azzz;
    bool inf = true;
    if (cnt > HEAVY_CHUNK) cnt = HEAVY_CHUNK;   // the rest of an over-full bucket is folded by k_accumulate_heavy
    for (u32 e = 0; e < cnt; e++) {
        const u32 code = srt[off + e];
        const u64* pp = pts + (size_t)24 * (code & 0x7fffffffu) + (par ? 6 : 0);
        FqU qx = fqu_unpack(fp_load<FqParams>(pp)), qy = fqu_unpack(fp_load<FqParams>(pp + 12));
        if (code & 0x80000000u) {
#pragma unroll
            for (int i = 0; i < 14; i++) qy.l[i] = fqu_4p(i) - qy.l[i];
            if (inf) qy = fqu_normalize(qy);
        }
        if (inf) {
            ax = qx;
            ay = qy;
            azz = fqu_one();
#pragma unroll
            for (int i = 0; i < 14; i++) azz.l[i] = par ? 0u : azz.l[i];
            azzz = azz;
            inf = false;
            continue;
        }
        if (!xyzzu2_acc_mixed(ax, ay, azz, azzz, qx, qy)) {
            u32 slot = par ? 0u : atomicAdd(exc_count, 1u);
            const u32 other = pair_swap_u32(slot);
            if (par) slot = other;
            if (slot < exc_cap) {
                if (!par) {
                    exc_list[3 * slot] = lane;
                    exc_list[3 * slot + 1] = (u32)b;
                    exc_list[3 * slot + 2] = code;
                }
                continue;
            }
            if (!par) dirty[(size_t)lane * B + b] = 1;
            return;
        }
    }
    u64* slot = buckets + (size_t)48 * ((size_t)lane * B + b) + (par ? 6 : 0);
    if (inf) {
        const Fq z = Fq::zero();
        fp_store<FqParams>(slot, z);
        fp_store<FqParams>(slot + 12, z);
        fp_store<FqParams>(slot + 24, z);
        fp_store<FqParams>(slot + 36, z);
        return;
    }
    fp_store<FqParams>(slot, fqu_pack(ax));
    fp_store<FqParams>(slot + 12, fqu_pack(ay));
    fp_store<FqParams>(slot + 24, fqu_pack(azz));
    fp_store<FqParams>(slot + 36, fqu_pack(azzz));
}
#endif

__global__ __launch_bounds__(128) CZK_FIX_ATTR void k_accumulate_u2_fix(const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B,
                                                          size_t sorted_stride, u64* buckets, const uint8_t* dirty, int ubuckets) {
    const unsigned lane = blockIdx.y;
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (size_t)gridDim.x * blockDim.x) {
    if (!dirty[(size_t)lane * B + b]) continue;
    u32 off = offsets[(size_t)lane * B + b], cnt = counts[(size_t)lane * B + b];
    Fq2 ax = Fq2::one(), ay = Fq2::one(), azz = Fq2::zero(), azzz = Fq2::zero();
    for (u32 e = 0; e < cnt; e++) {
        u32 code = srt[off + e];
        const u64* pp = pts + (size_t)24 * (code & 0x7fffffffu);
        Fq2 qx = fq2_from_table_u(pp), qy = fq2_from_table_u(pp + 12);
        if (code & 0x80000000u) qy = f_neg(qy);
        xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy);
    }
    bucket_store_sat<Fq2>(buckets + (size_t)48 * ((size_t)lane * B + b), XYZZ<Fq2>{ax, ay, azz, azzz}, ubuckets);
    }
}

__global__ void k_accumulate_u2_cleanup(const u64* pts, size_t B, u64* buckets, const uint8_t* dirty, const u32* exc_count, const u32* exc_list,
                                        u32 exc_cap, int ubuckets) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    u32 n = *exc_count;
    if (n > exc_cap) n = exc_cap;
    for (u32 k = 0; k < n; k++) {
        u32 lane = exc_list[3 * k], b = exc_list[3 * k + 1], code = exc_list[3 * k + 2];
        if (dirty[(size_t)lane * B + b]) continue;
        u64* slot = buckets + (size_t)48 * ((size_t)lane * B + b);
        XYZZ<Fq2> acc = bucket_load_sat<Fq2>(slot, ubuckets);
        const u64* pp = pts + (size_t)24 * (code & 0x7fffffffu);
        Fq2 qx = fq2_from_table_u(pp), qy = fq2_from_table_u(pp + 12);
        if (code & 0x80000000u) qy = f_neg(qy);
        xyzz_acc_mixed(acc.x, acc.y, acc.zz, acc.zzz, qx, qy);
        bucket_store_sat<Fq2>(slot, acc, ubuckets);
    }
}

#endif

}  // namespace czk

// msm.hip -- variable-base multi-scalar multiplication over BLS12-377 G1 / G2 for gfx950.
//
// Replaces VariableBaseMSM::multi_scalar_mul (algebra/ec/src/msm/variable_base.rs:12-106) and
// AffineCurve::multi_scalar_mul (algebra/ec/src/lib.rs:300-311) as reached from the MPC wrappers
// (mpc-algebra/src/wire/pairing.rs:746-809, share/spdz.rs:440-446).  The result is the same group element
// sum_i s_i * P_i; Jacobian representatives differ from the reference's (they are not canonical), so parity is
// checked in affine form, as the reference's own MSM test does (algebra/test-templates/src/msm.rs:16-33).
//
// MI355X-first design -- NOT the reference's serial "for each window: fill 2^c-1 buckets, running-sum them,
// then Horner over windows with c doublings":
//   * HBM capacity is traded for arithmetic.  The bases are public and reused across proofs
//     (groth16/src/data_structures.rs:132-149), so registration stores 2^(c*w) * P_i for every window w
//     (W x the table; ~9 GB for a 2^20-constraint Groth16 key, out of 288 GB).  Every (scalar, window)
//     digit then lands in ONE shared bucket set: a single bucket reduction per MSM and no window-combine
//     doubling chain (256 dependent doublings in the reference) at all.
//   * signed digits: 2^(c-1) buckets of weight 1..2^(c-1); a negative digit adds -P (y -> p - y).
//   * digits are counting-sorted by bucket (partitioned: per-block LDS histograms, 1024-bucket partitions, one
//     workgroup per partition), then one thread owns one bucket and folds its points with the mixed addition (same
//     edge cases as short_weierstrass_jacobian.rs:570-597) -- no atomics or locks on group elements; buckets are
//     visited in descending-population order so the 64 lanes of a wave do equal work; a bucket with more than 1024
//     entries is cut into work items (msm_acc.h, over-full buckets).
//   * buckets live in XYZZ coordinates (curve.h): mixed addition 8M + 2S instead of 7M + 4S, addition 12M + 2S
//     instead of 11M + 5S; one conversion back to the reference's Jacobian triple per result.
//   * bucket reduction sum_b (b+1) * B_b: multi-level chunked running sums (the reference's :82-86 running
//     sum, applied per chunk, with the chunk offsets folded in at the next level) down to 1024 entries per lane,
//     then bit-sum tree reductions (msm_acc.h k_reduce_tail_*).
//   * consecutive MSMs pipeline over three internal streams (msm_enqueue below).
//   * `lanes` scalar vectors that share the bases (SPDZ sh / mac lanes) ride on gridDim.y.
//   * bases registered WITHOUT tables (CZK_MEM_NO_TABLES; one-shot callers): the W digit windows of a lane become W virtual lanes
//     of the same kernels, one bucket set each, and the per-window results are combined on the host (msm_enqueue, b->split).
// All arithmetic is 32-bit-limb integer VALU (field.h); nothing here is MFMA-shaped.
#include <stdlib.h>
#include <string.h>

#include "czk_internal.h"

namespace czk {

// curves/bls12_377/src/curves/g1.rs:46-51, g2.rs:64-86 generators, Montgomery form, 32-bit limbs
__device__ __forceinline__ Affine<Fq> generator(Fq*) {
    const u32 gx[12] = {0x772451f4u, 0x260f33b9u, 0x169d5658u, 0xc54dd773u, 0x69a510ddu, 0x5c1551c4u,
                        0x425e1698u, 0x761662e4u, 0x6f065272u, 0xc97d78ccu, 0xb361fd4du, 0x00a41206u};
    const u32 gy[12] = {0xb8cb81f3u, 0x8193961fu, 0x5f44adb8u, 0x00638d4cu, 0xd4daf54au, 0xfafaf3dau,
                        0xd655cd18u, 0xc27849e2u, 0x01d52814u, 0x2ec3ddb4u, 0x26303c71u, 0x007da933u};
    Affine<Fq> g;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        g.x.l[i] = gx[i];
        g.y.l[i] = gy[i];
    }
    return g;
}
__device__ __forceinline__ Affine<Fq2> generator(Fq2*) {
    const u32 x0[12] = {0xf268725bu, 0x68904082u, 0x4f45328bu, 0x668f2ea7u, 0x802be84fu, 0xebca7a65u,
                        0xc1ada3e6u, 0x1e1850f4u, 0x588ef1e9u, 0x830dc22du, 0x767c0982u, 0x01862a81u};
    const u32 x1[12] = {0xc91c7f39u, 0x5f02a915u, 0x388da2a7u, 0xf8c553bau, 0xbd198850u, 0xd51a416du,
                        0x8ae3073au, 0xe943c6f3u, 0x259a4981u, 0xffe24aa8u, 0x1e73dfddu, 0x01185339u};
    const u32 y0[12] = {0x7881430fu, 0xd5b19b89u, 0xa5b371edu, 0x05be9118u, 0x86c131eeu, 0x6063f91fu,
                        0xe8f4ec19u, 0x3244a61bu, 0x9f9a3a12u, 0xa02e425bu, 0x4f3360d2u, 0x018af8c0u};
    const u32 y1[12] = {0x1a5b96f5u, 0x57601ac7u, 0x14f2440eu, 0xe99acc17u, 0x10118ea9u, 0x2339612fu,
                        0x3b1cd722u, 0x8321e68au, 0x0cc74917u, 0x2b543b05u, 0xb396c112u, 0x00590182u};
    Affine<Fq2> g;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        g.x.c0.l[i] = x0[i];
        g.x.c1.l[i] = x1[i];
        g.y.c0.l[i] = y0[i];
        g.y.c1.l[i] = y1[i];
    }
    return g;
}

// ------------------------------------------------------------------------------------------------
// setup kernels: fixed-base points, window multiples, batched Jacobian -> affine
// ------------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(128) void k_fixed_base(const u64* k, size_t n, u64* out_jac) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine<F> g = generator((F*)nullptr);
    Jac<F> acc = Jac<F>::zero();
    for (int limb = 3; limb >= 0; limb--) {
        u64 w = k[4 * i + limb];
        for (int b = 63; b >= 0; b--) {
            acc = jac_double(acc);
            if ((w >> b) & 1) acc = jac_add_mixed(acc, g, false);
        }
    }
    jac_store<F>(out_jac + (size_t)GT<F>::JW * i, acc);
}

// out = 2^c * in   (in affine + infinity flag, out Jacobian)
template <class F>
__global__ __launch_bounds__(128) void k_dbl_c(const u64* aff, const uint8_t* inf, size_t n, unsigned c, u64* out_jac) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jac<F> p;
    if (inf[i]) {
        p = Jac<F>::zero();
    } else {
        Affine<F> a = aff_load<F>(aff + (size_t)GT<F>::AW * i);
        p = Jac<F>{a.x, a.y, F::one()};
        for (unsigned k = 0; k < c; k++) p = jac_double(p);
    }
    jac_store<F>(out_jac + (size_t)GT<F>::JW * i, p);
}

// Montgomery's trick over CH consecutive points per thread (one field inversion per CH points).
// scratch: n field elements.
template <class F>
__global__ __launch_bounds__(128) void k_batch_to_affine(const u64* jac, size_t n, unsigned CH, u64* scratch, u64* out_aff,
                                                        uint8_t* out_inf) {
    constexpr int JW = GT<F>::JW, AW = GT<F>::AW, FW = GT<F>::FW;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = t * CH;
    if (start >= n) return;
    size_t end = start + CH < n ? start + CH : n;
    F acc = F::one();
    for (size_t i = start; i < end; i++) {
        F z = FieldIO<F>::load(jac + JW * i + 2 * FW);
        FieldIO<F>::store(scratch + FW * i, acc);
        if (!z.is_zero()) acc = f_mul(acc, z);
    }
    F inv = f_inv(acc);
    for (size_t i = end; i-- > start;) {
        F z = FieldIO<F>::load(jac + JW * i + 2 * FW);
        Affine<F> a;
        if (z.is_zero()) {
            a.x = F::zero();
            a.y = F::one();
            out_inf[i] = 1;
        } else {
            F zinv = f_mul(inv, FieldIO<F>::load(scratch + FW * i));
            inv = f_mul(inv, z);
            F zi2 = f_sqr(zinv);
            a.x = f_mul(FieldIO<F>::load(jac + JW * i), zi2);
            a.y = f_mul(FieldIO<F>::load(jac + JW * i + FW), f_mul(zi2, zinv));
            out_inf[i] = 0;
        }
        aff_store<F>(out_aff + (size_t)AW * i, a);
    }
}

// ------------------------------------------------------------------------------------------------
// digit extraction + counting sort
// ------------------------------------------------------------------------------------------------
// digits[(lane*W + w)*size + i] = 0 (skip) or |d| | sign<<31, d in [-2^(c-1), 2^(c-1)].
__global__ void k_digits(const u64* scalars, size_t n_scalars, size_t size, int montgomery, unsigned c, unsigned W,
                         const uint8_t* inf, size_t n_bases, u32* digits, u32* ranks, u32* counts, size_t B) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    const unsigned lane = blockIdx.y;
    Fr s = fp_load<FrParams>(scalars + 4 * ((size_t)lane * n_scalars + i));
    if (montgomery) s = fp_into_repr(s);   // ec/src/lib.rs:305-307
    u32 carry = 0;
    const unsigned W_hi = msm_full_windows(c);
    for (unsigned w = 0; w < W; w++) {
        const unsigned bit = msm_win_bit(c, W_hi, w), cw = msm_win_width(c, W_hi, w);
        const u32 half = 1u << (cw - 1);
        u32 v = 0;
        if (bit < 256) {
            unsigned limb = bit >> 5, off = bit & 31;
            u64 two = (u64)s.l[limb] | ((limb + 1 < 8) ? ((u64)s.l[limb + 1] << 32) : 0);
            v = (u32)(two >> off) & ((1u << cw) - 1u);
        }
        v += carry;
        u32 code;
        if (v > half) {
            code = ((1u << cw) - v) | 0x80000000u;
            carry = 1;
        } else {
            code = v;
            carry = 0;
        }
        if (inf[(size_t)w * n_bases + i]) code = 0;   // add_assign_mixed skips infinity (short_weierstrass_jacobian.rs:571-573)
        if ((code & 0x7fffffffu) == 0) code = 0;
        digits[((size_t)lane * W + w) * size + i] = code;
        // the histogram atomic also hands out the entry's rank inside its bucket, so the scatter needs no second atomic
        if (code) ranks[((size_t)lane * W + w) * size + i] = atomicAdd(&counts[(size_t)lane * B + (code & 0x7fffffffu) - 1], 1u);
    }
}

// exclusive scan of counts[lane][0..B) -> offsets.  Three phases:
// per-tile sums (tile = 2048 entries), scan of the tile sums (one block per lane), per-tile exclusive scan.
constexpr unsigned SCAN_TILE = 2048;
__global__ __launch_bounds__(256) void k_scan_tile_sums(const u32* counts, size_t B, u32* tile_sums, size_t n_tiles) {
    __shared__ u32 red[256];
    const size_t tile = blockIdx.x;
    const u32* cnt = counts + (size_t)blockIdx.y * B + tile * SCAN_TILE;
    size_t lim = B - tile * SCAN_TILE < SCAN_TILE ? B - tile * SCAN_TILE : SCAN_TILE;
    u32 s = 0;
    for (unsigned i = threadIdx.x; i < lim; i += 256) s += cnt[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (unsigned d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[(size_t)blockIdx.y * n_tiles + tile] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_tiles(u32* tile_sums, size_t n_tiles) {
    __shared__ u32 part[1024];
    u32* ts = tile_sums + (size_t)blockIdx.x * n_tiles;
    const unsigned tid = threadIdx.x;
    size_t per = (n_tiles + 1023) / 1024;
    size_t start = tid * per, end = start + per < n_tiles ? start + per : n_tiles;
    u32 sum = 0;
    for (size_t i = start; i < end; i++) sum += ts[i];
    part[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (size_t i = start; i < end; i++) {
        u32 v = ts[i];
        ts[i] = run;
        run += v;
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(u32* counts, u32* offsets, size_t B, const u32* tile_sums, size_t n_tiles) {
    __shared__ u32 part[256];
    const size_t tile = blockIdx.x;
    u32* cnt = counts + (size_t)blockIdx.y * B + tile * SCAN_TILE;
    u32* off = offsets + (size_t)blockIdx.y * B + tile * SCAN_TILE;
    size_t lim = B - tile * SCAN_TILE < SCAN_TILE ? B - tile * SCAN_TILE : SCAN_TILE;
    const unsigned tid = threadIdx.x;
    constexpr unsigned PER = SCAN_TILE / 256;
    u32 v[PER];
    u32 s = 0;
#pragma unroll
    for (unsigned k = 0; k < PER; k++) {
        unsigned i = tid * PER + k;
        v[k] = i < lim ? cnt[i] : 0;
        s += v[k];
    }
    part[tid] = s;
    __syncthreads();
    for (unsigned d = 1; d < 256; d <<= 1) {
        u32 x = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    u32 run = tile_sums[(size_t)blockIdx.y * n_tiles + tile] + (tid ? part[tid - 1] : 0);
#pragma unroll
    for (unsigned k = 0; k < PER; k++) {
        unsigned i = tid * PER + k;
        if (i < lim) off[i] = run;
        run += v[k];
    }
}

// sorted[lane][offsets[b] + k] = (w * n_bases + i) | sign<<31
__global__ void k_scatter(const u32* digits, const u32* ranks, size_t size, unsigned W, size_t n_bases, const u32* offsets, size_t B,
                          u32* sorted) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)W * size) return;
    const unsigned lane = blockIdx.y;
    u32 code = digits[(size_t)lane * W * size + e];
    if (!code) return;
    size_t w = e / size, i = e - w * size;
    size_t b = (code & 0x7fffffffu) - 1;
    u32 pos = offsets[(size_t)lane * B + b] + ranks[(size_t)lane * W * size + e];
    sorted[(size_t)lane * W * size + pos] = (u32)(w * n_bases + i) | (code & 0x80000000u);
}

// ------------------------------------------------------------------------------------------------
// partitioned counting sort (default).  The one-pass sort above costs one global atomic and one random 4-byte store per
// entry -- 54 M of each per 4-lane 2^20-point MSM -- and those are what slows the accumulate kernels of the neighbouring
// MSMs in the pipeline (measured: every ms of k_scatter overlap inflates an accumulate kernel by ~0.6 ms).  Here the
// buckets are grouped into partitions of 1024 (by the LOW bits of the bucket index, so the few thousand over-full buckets of
// the 13-bit top window spread over all partitions; `sorted` is partition-major, which the accumulate kernel does not care about): (1) digits + per-block LDS histogram of partitions (global atomics: one per
// block and partition); (2) scan of the partition sizes; (3) entries move into their partition's region -- per block, the
// entries of one partition land in one contiguous run; (4) one workgroup per partition counts, scans and places its
// entries with LDS atomics and writes the partition's slice of `sorted`, `offsets` and `counts`.
// ------------------------------------------------------------------------------------------------
constexpr unsigned PART_LOG = 10, PART_BUCKETS = 1u << PART_LOG, MAX_PARTS = 2048;

// split != 0 (bases without window tables): every window is its own bucket set, i.e. window w of lane l is virtual lane l W + w of
// everything downstream (the `digits` layout is the same either way); partition counts are then kept per window.
__global__ __launch_bounds__(256) void k_digits_part(const u64* scalars, size_t n_scalars, size_t size, int montgomery, unsigned c, unsigned W,
                                                     const uint8_t* inf, size_t n_bases, u32* digits, u32* part_counts, unsigned n_parts, int split) {
    __shared__ u32 h[MAX_PARTS];
    const unsigned n_hist = split ? W * n_parts : n_parts;
    for (unsigned t = threadIdx.x; t < n_hist; t += 256) h[t] = 0;
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = blockIdx.y;
    if (i < size) {
        Fr s = fp_load<FrParams>(scalars + 4 * ((size_t)lane * n_scalars + i));
        if (montgomery) s = fp_into_repr(s);   // ec/src/lib.rs:305-307
        u32 carry = 0;
        const unsigned W_hi = msm_full_windows(c);
        for (unsigned w = 0; w < W; w++) {
            const unsigned bit = msm_win_bit(c, W_hi, w), cw = msm_win_width(c, W_hi, w);
            const u32 half = 1u << (cw - 1);
            u32 v = 0;
            if (bit < 256) {
                unsigned limb = bit >> 5, off = bit & 31;
                u64 two = (u64)s.l[limb] | ((limb + 1 < 8) ? ((u64)s.l[limb + 1] << 32) : 0);
                v = (u32)(two >> off) & ((1u << cw) - 1u);
            }
            v += carry;
            u32 code;
            if (v > half) {
                code = ((1u << cw) - v) | 0x80000000u;
                carry = 1;
            } else {
                code = v;
                carry = 0;
            }
            if (inf[(split ? 0 : (size_t)w * n_bases) + i]) code = 0;   // add_assign_mixed skips infinity (short_weierstrass_jacobian.rs:571-573)
            if ((code & 0x7fffffffu) == 0) code = 0;
            digits[((size_t)lane * W + w) * size + i] = code;
            if (code) atomicAdd(&h[(split ? w * n_parts : 0) + (((code & 0x7fffffffu) - 1) & (n_parts - 1))], 1u);
        }
    }
    __syncthreads();
    for (unsigned t = threadIdx.x; t < n_hist; t += 256)
        if (h[t]) atomicAdd(&part_counts[(size_t)lane * n_hist + t], h[t]);   // split: (lane W + w) n_parts + part
}
// part_base[lane][0 .. n_parts] = exclusive scan of part_counts; cursors cleared.  One block per lane.
__global__ __launch_bounds__(1024) void k_part_scan(const u32* part_counts, u32* part_base, u32* part_cursor, unsigned n_parts) {
    __shared__ u32 part[1024];
    const unsigned lane = blockIdx.x, tid = threadIdx.x;
    const u32* cnt = part_counts + (size_t)lane * n_parts;
    u32* base = part_base + (size_t)lane * (n_parts + 1);
    const unsigned per = (n_parts + 1023) / 1024;
    unsigned start = tid * per, end = start + per < n_parts ? start + per : n_parts;
    u32 sum = 0;
    for (unsigned i = start; i < end; i++) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
        u32 v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = tid ? part[tid - 1] : 0;
    for (unsigned i = start; i < end; i++) {
        base[i] = run;
        part_cursor[(size_t)lane * n_parts + i] = 0;
        run += cnt[i];
    }
    if (tid == 1023) base[n_parts] = part[1023];
}
// entries -> partition regions: part_idx[dst] = point code, part_lb[dst] = bucket index inside the partition
#ifndef CZK_PS_TILE
#define CZK_PS_TILE 16
#endif
constexpr unsigned PS_TILE = CZK_PS_TILE;   // entries per thread
// NT threads x PS_TILE entries per tile.  A tile leaves as one run per partition, so a run is tile / n_parts entries long: 256 threads (8-entry
// runs at 512 partitions) for the common case, 1024 threads for calls with 2048 partitions (3 * 2^20-point commitments, 2^22-point queries), whose
// runs would otherwise be 2 entries: partial sectors again (same-box: Marlin 5.67 -> 6.01 proofs/s, Groth16 at 2^22 3.12 -> 3.28; at 1024 partitions --
// the 2^21-point h query, Plonk's 3 * 2^19-point commitments -- and at 512 (2^20 points) 512 threads: the isolated sort of 2^21 x 4 lanes 2.74 -> 1.73 ms,
// of 2^20 x 4 lanes 1.03 -> 0.87 ms, per proof within noise).
static size_t part_scatter_lds(unsigned nt, unsigned n_parts) { return (size_t)(3 * n_parts + nt) * 4 + (size_t)nt * PS_TILE * (4 + 2 + 2); }
template <unsigned NT>
__global__ __launch_bounds__(NT) void k_part_scatter(const u32* digits, size_t size, unsigned W, size_t n_bases, const u32* part_base, u32* part_cursor,
                                                     unsigned n_parts, unsigned part_shift, u32* part_idx, uint16_t* part_lb) {
    // The tile is ordered by partition in LDS first and leaves as runs: consecutive lanes then write consecutive addresses of a partition's region
    // (a store instruction touches ~8 sectors instead of 64 partial ones).
    extern __shared__ u32 pscat_lds[];
    u32 *h = pscat_lds, *base = h + n_parts, *lofs = base + n_parts, *red = lofs + n_parts, *st_idx = red + NT;
    uint16_t *st_lb = (uint16_t*)(st_idx + NT * PS_TILE), *st_pt = st_lb + NT * PS_TILE;
    for (unsigned t = threadIdx.x; t < n_parts; t += NT) h[t] = 0;
    __syncthreads();
    const unsigned lane = blockIdx.y, tid = threadIdx.x;
    const size_t total = (size_t)W * size, tile0 = (size_t)blockIdx.x * NT * PS_TILE;
    u32 code[PS_TILE], rank[PS_TILE];
#pragma unroll
    for (unsigned k = 0; k < PS_TILE; k++) {
        size_t e = tile0 + (size_t)k * NT + tid;
        code[k] = e < total ? digits[(size_t)lane * total + e] : 0u;
        if (code[k]) rank[k] = atomicAdd(&h[((code[k] & 0x7fffffffu) - 1) & (n_parts - 1)], 1u);
    }
    __syncthreads();
    // exclusive scan of h over the partitions (n_parts <= 2048: up to 8 per thread) -> lofs; global bases
    const unsigned per = (n_parts + NT - 1) / NT;
    u32 sum = 0;
    for (unsigned i = tid * per; i < tid * per + per && i < n_parts; i++) sum += h[i];
    red[tid] = sum;
    __syncthreads();
    for (unsigned d = 1; d < NT; d <<= 1) {
        u32 x = tid >= d ? red[tid - d] : 0;
        __syncthreads();
        red[tid] += x;
        __syncthreads();
    }
    u32 run = tid ? red[tid - 1] : 0;
    for (unsigned i = tid * per; i < tid * per + per && i < n_parts; i++) {
        lofs[i] = run;
        run += h[i];
        if (h[i]) base[i] = part_base[(size_t)lane * (n_parts + 1) + i] + atomicAdd(&part_cursor[(size_t)lane * n_parts + i], h[i]);
    }
    const u32 n_tile = red[NT - 1];
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < PS_TILE; k++) {
        if (!code[k]) continue;
        size_t e = tile0 + (size_t)k * NT + tid;
        size_t w = e / size, i = e - w * size;
        u32 b = (code[k] & 0x7fffffffu) - 1;
        const u32 pt = b & (n_parts - 1), slot = lofs[pt] + rank[k];
        st_idx[slot] = (u32)(w * n_bases + i) | (code[k] & 0x80000000u);
        st_lb[slot] = (uint16_t)(b >> part_shift);
        st_pt[slot] = (uint16_t)pt;
    }
    __syncthreads();
    for (u32 sl = tid; sl < n_tile; sl += NT) {
        const u32 pt = st_pt[sl];
        const size_t dst = (size_t)lane * total + base[pt] + (sl - lofs[pt]);
        part_idx[dst] = st_idx[sl];
        part_lb[dst] = st_lb[sl];
    }
}
// one workgroup per (partition, lane): bucket counts, offsets and the final placement of the partition's entries.  The placement is staged
// in LDS when the partition fits (`cap` entries of dynamic LDS behind the counters): 4-byte stores to random positions of the partition's
// output cost a sector write each, the staged copy leaves as coalesced 4 KiB rows.  Larger partitions (the 2^21-point `h` query: 53 k
// entries) place directly, as rounds 1 - 3 did for every partition.
constexpr unsigned PSORT_THREADS = 1024;
__global__ __launch_bounds__(PSORT_THREADS) void k_part_sort(const u32* part_idx, const uint16_t* part_lb, const u32* part_base, unsigned n_parts, unsigned part_shift,
                                                             size_t total, size_t B, u32* sorted, u32* offsets, u32* counts, u32 cap) {
    extern __shared__ u32 psort_lds[];
    u32 *cnt = psort_lds, *cur = psort_lds + PART_BUCKETS, *red = psort_lds + 2 * PART_BUCKETS, *stage = psort_lds + 2 * PART_BUCKETS + PSORT_THREADS;
    const unsigned p = blockIdx.x, lane = blockIdx.y, tid = threadIdx.x;
    const u32 r0 = part_base[(size_t)lane * (n_parts + 1) + p], r1 = part_base[(size_t)lane * (n_parts + 1) + p + 1];
    for (unsigned t = tid; t < PART_BUCKETS; t += PSORT_THREADS) cnt[t] = 0;
    __syncthreads();
    const uint16_t* lb = part_lb + (size_t)lane * total;
    const u32* idx = part_idx + (size_t)lane * total;
    // four entries per thread and iteration, loads issued together
    for (u32 j = r0 + tid; j < r1; j += 4 * PSORT_THREADS) {
        const bool h1 = j + PSORT_THREADS < r1, h2 = j + 2 * PSORT_THREADS < r1, h3 = j + 3 * PSORT_THREADS < r1;
        const uint16_t l0 = lb[j], l1 = h1 ? lb[j + PSORT_THREADS] : (uint16_t)0, l2 = h2 ? lb[j + 2 * PSORT_THREADS] : (uint16_t)0,
                       l3 = h3 ? lb[j + 3 * PSORT_THREADS] : (uint16_t)0;
        atomicAdd(&cnt[l0], 1u);
        if (h1) atomicAdd(&cnt[l1], 1u);
        if (h2) atomicAdd(&cnt[l2], 1u);
        if (h3) atomicAdd(&cnt[l3], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..1024): one entry per thread
    const u32 v = tid < PART_BUCKETS ? cnt[tid] : 0u;
    red[tid] = v;
    __syncthreads();
    for (unsigned d = 1; d < PSORT_THREADS; d <<= 1) {
        u32 x = tid >= d ? red[tid - d] : 0;
        __syncthreads();
        red[tid] += x;
        __syncthreads();
    }
    if (tid < PART_BUCKETS) {
        const u32 run = red[tid] - v;                      // exclusive prefix, relative to the partition
        size_t b = ((size_t)tid << part_shift) | p;          // bucket = (index inside the partition, partition)
        cur[tid] = run;
        if (b < B) {
            offsets[(size_t)lane * B + b] = r0 + run;
            counts[(size_t)lane * B + b] = v;
        }
    }
    __syncthreads();
    u32* out = sorted + (size_t)lane * total;
    const bool staged = r1 - r0 <= cap;
    for (u32 j = r0 + tid; j < r1; j += 4 * PSORT_THREADS) {
        const bool h1 = j + PSORT_THREADS < r1, h2 = j + 2 * PSORT_THREADS < r1, h3 = j + 3 * PSORT_THREADS < r1;
        const uint16_t l0 = lb[j], l1 = h1 ? lb[j + PSORT_THREADS] : (uint16_t)0, l2 = h2 ? lb[j + 2 * PSORT_THREADS] : (uint16_t)0,
                       l3 = h3 ? lb[j + 3 * PSORT_THREADS] : (uint16_t)0;
        const u32 v0 = idx[j], v1 = h1 ? idx[j + PSORT_THREADS] : 0u, v2 = h2 ? idx[j + 2 * PSORT_THREADS] : 0u, v3 = h3 ? idx[j + 3 * PSORT_THREADS] : 0u;
        if (staged) {
            stage[atomicAdd(&cur[l0], 1u)] = v0;
            if (h1) stage[atomicAdd(&cur[l1], 1u)] = v1;
            if (h2) stage[atomicAdd(&cur[l2], 1u)] = v2;
            if (h3) stage[atomicAdd(&cur[l3], 1u)] = v3;
        } else {
            out[r0 + atomicAdd(&cur[l0], 1u)] = v0;
            if (h1) out[r0 + atomicAdd(&cur[l1], 1u)] = v1;
            if (h2) out[r0 + atomicAdd(&cur[l2], 1u)] = v2;
            if (h3) out[r0 + atomicAdd(&cur[l3], 1u)] = v3;
        }
    }
    if (staged) {
        __syncthreads();
        for (u32 k = tid; k < r1 - r0; k += PSORT_THREADS) out[r0 + k] = stage[k];
    }
}

// ------------------------------------------------------------------------------------------------
// load balancing: order the buckets by population (descending) so the 64 lanes of a wave fold the same number
// of points (bucket sizes are ~Poisson: without this a wave waits for its fullest bucket, ~30% of the time)
// ------------------------------------------------------------------------------------------------
constexpr unsigned CNT_BINS = 2048;
__global__ __launch_bounds__(1024) void k_count_hist(const u32* counts, size_t B, u32* hist) {
    __shared__ u32 h[CNT_BINS];   // block-private histogram: populations cluster on a few values
    for (unsigned i = threadIdx.x; i < CNT_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        u32 c = counts[(size_t)blockIdx.y * B + b];
        atomicAdd(&h[c < CNT_BINS ? c : CNT_BINS - 1], 1u);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < CNT_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[(size_t)blockIdx.y * CNT_BINS + i], h[i]);
}
// start[v] = number of buckets with a larger population; one block of CNT_BINS/2 threads per lane
__global__ __launch_bounds__(1024) void k_count_starts(u32* hist) {
    __shared__ u32 part[CNT_BINS];
    u32* h = hist + (size_t)blockIdx.x * CNT_BINS;
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < CNT_BINS; i += blockDim.x) part[i] = h[CNT_BINS - 1 - i];   // reversed: descending order
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (unsigned i = 0; i < CNT_BINS; i++) {
            u32 c = part[i];
            part[i] = run;
            run += c;
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < CNT_BINS; i += blockDim.x) h[CNT_BINS - 1 - i] = part[i];
}
__global__ __launch_bounds__(1024) void k_count_scatter(const u32* counts, size_t B, u32* starts, u32* perm) {
    __shared__ u32 h[CNT_BINS], base[CNT_BINS];
    for (unsigned i = threadIdx.x; i < CNT_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 bin = 0, rank = 0;
    if (b < B) {
        u32 c = counts[(size_t)blockIdx.y * B + b];
        bin = c < CNT_BINS ? c : CNT_BINS - 1;
        rank = atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < CNT_BINS; i += blockDim.x)
        if (h[i]) base[i] = atomicAdd(&starts[(size_t)blockIdx.y * CNT_BINS + i], h[i]);
    __syncthreads();
    if (b < B) perm[(size_t)blockIdx.y * B + base[bin] + rank] = (u32)b;
}

// ------------------------------------------------------------------------------------------------
// host drivers
// ------------------------------------------------------------------------------------------------
struct Bump {
    char* base;
    size_t off = 0;
    template <class T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = (T*)(base + off);
        off += count * sizeof(T);
        return p;
    }
};

static unsigned num_windows(unsigned c) { return msm_num_windows(c); }

// window width for n bases: minimise W(c) * n mixed additions + ~3 * 2^(c-1) reduction additions.  (In instruction terms
// a bucket costs ~6 mixed additions to reduce, but weighting it so -- c = 17 at n = 2^20 -- lengthens the accumulate kernels,
// which are the critical stream of the pipeline: measured 98 -> 102 ms per proof.)  Widths whose TOP window is only a few
// bits wide are skipped for large n: its digits pile n / 2^t points onto each of ~2^t buckets, and a bucket is one thread's
// serial chain (c = 19: 7 top bits -> 37 buckets of 28 k points at n = 2^20).
static unsigned choose_c(size_t n) {
    unsigned best = 2;
    double best_cost = 1e300;
    for (unsigned c = 2; c <= 22; c++) {
        const unsigned W = num_windows(c), top_bits = 254 - (W - 1) * c;
        if (n >= 16384 && top_bits < 10) continue;
        double cost = (double)W * (double)(n ? n : 1) + 3.0 * (double)((size_t)1 << (c - 1));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// Without window tables every window has its own bucket set: W(c) * n mixed additions + W(c) bucket reductions of 2^(c-1) buckets
// (~4 mixed additions' worth of instructions per bucket).  The per-window partition histograms of k_digits_part must fit its LDS array.
static unsigned choose_c_split(size_t n) {
    unsigned best = 2;
    double best_cost = 1e300;
    for (unsigned c = 2; c <= 20; c++) {
        const unsigned W = num_windows(c), top_bits = 254 - (W - 1) * c;
        const size_t B = (size_t)1 << (c - 1), n_parts = (B + PART_BUCKETS - 1) >> PART_LOG;
        if ((size_t)W * n_parts > MAX_PARTS) continue;
        if (n >= 16384 && top_bits < 10) continue;   // a narrow top window piles n / 2^bits points on each of its few buckets
        double cost = (double)W * (double)(n ? n : 1) + 4.0 * (double)W * (double)B;
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// G1 in twisted Edwards form (te.h): a saturated short-Weierstrass table of `count` points -> a freshly allocated niels table
// (count x 24 u64: three coordinates of 14 x 28-bit limbs in 16 u32 each).  *ok = false (and no table) when some point has no image under the map -- such a point has even order
// and is never an element of G1; the caller then keeps the XYZZ path, which is complete on all of E.
static int te_table_from_sw(czk_ctx* ctx, const u64* sw, const uint8_t* inf, size_t count, u64** out, bool* ok) {
    *out = nullptr;
    *ok = false;
    u64 *te = nullptr, *scr = nullptr;
    u32* bad = nullptr;
    hipError_t e = hipMalloc(&te, count * 24 * 8);
    if (e == hipSuccess) e = hipMalloc(&scr, count * 6 * 8);
    if (e == hipSuccess) e = hipMalloc(&bad, 4);
    if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, ctx->stream);
    u32 hbad = 1;
    if (e == hipSuccess) {
        launch_sw_to_te_niels(ctx->stream, sw, inf, count, scr, te, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (scr) (void)hipFree(scr);
    if (bad) (void)hipFree(bad);
    if (e != hipSuccess || hbad) {
        if (te) (void)hipFree(te);
        if (e != hipSuccess) return set_err(ctx, e == hipErrorOutOfMemory ? CZK_ERR_NOMEM : CZK_ERR_HIP, std::string("twisted Edwards table: ") + hipGetErrorString(e));
        return CZK_OK;
    }
    *out = te;
    *ok = true;
    return CZK_OK;
}

// ------------------------------------------------------------------------------------------------
// Subgroup membership of registered bases: the reference's `is_in_correct_subgroup_assuming_on_curve` is `self.mul(r).is_zero()`
// (short_weierstrass_jacobian.rs:131), enforced when a key is deserialised (:868, :881); its MSM itself is complete on every curve point.
// The twisted Edwards G1 kernels are exception-free exactly on the prime-order subgroup, so a caller that cannot vouch for its bases asks
// for this check (czk_bases_check_subgroup, or CZK_MEM_CHECK_SUBGROUP at registration: a failing base keeps the handle on the XYZZ kernels).
// One thread per point: on-curve test (y^2 = x^3 + b), then [r] P by MSB-first double-and-add with the complete Jacobian formulas of curve.h.
// ------------------------------------------------------------------------------------------------
template <class F>
struct CurveB;
template <>
struct CurveB<Fq> {   // COEFF_B = 1 (curves/bls12_377/src/curves/g1.rs:23)
    static CZK_HD Fq get() { return Fq::one(); }
};
template <>
struct CurveB<Fq2> {   // COEFF_B = (0, 1552...4906) (curves/bls12_377/src/curves/g2.rs:28-34), c1 in Montgomery form
    static CZK_HD Fq2 get() {
        Fq2 b = Fq2::zero();
        constexpr u32 m[12] = {0x66666685u, 0x80722666u, 0x899999a9u, 0x8df55926u, 0xd64f34cfu, 0x7fe4561au,
                               0xb6e4f01bu, 0xb95da6d8u, 0xfc142743u, 0x4b747cccu, 0x70f49f43u, 0x0039c3fau};
#pragma unroll
        for (int i = 0; i < 12; i++) b.c1.l[i] = m[i];
        return b;
    }
};
template <class F>
__global__ __launch_bounds__(128) void k_subgroup_check(const u64* aff, const uint8_t* inf, size_t n, u32* bad) {
    // r = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001, 253 bits (curves/bls12_377/src/fields/fr.rs MODULUS)
    constexpr u32 R[8] = {0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (inf && inf[i])) return;
    const Affine<F> a = aff_load<F>(aff + (size_t)GT<F>::AW * i);
    bool ok = f_sqr(a.y) == f_add(f_mul(f_sqr(a.x), a.x), CurveB<F>::get());
    if (ok) {
        Jac<F> p{a.x, a.y, F::one()};
        for (int bit = 251; bit >= 0; bit--) {   // bit 252 is the leading one
            p = jac_double(p);
            if ((R[bit >> 5] >> (bit & 31)) & 1u) p = jac_add_mixed(p, a, false);
        }
        ok = p.is_zero();
    }
    if (!ok) atomicAdd(bad, 1u);
}
// pts: `n` affine points in the reference's (saturated Montgomery) form, device memory
template <class F>
static int subgroup_check_impl(czk_ctx* ctx, const u64* pts, const uint8_t* inf, size_t n, size_t* out_bad) {
    *out_bad = 0;
    if (!n) return CZK_OK;
    u32* bad = nullptr;
    u32 h = 0;
    CZK_HIP(ctx, hipMalloc(&bad, 4));
    hipError_t e = hipMemsetAsync(bad, 0, 4, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_subgroup_check<F>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, pts, inf, n, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(bad);
    if (e != hipSuccess) return set_err(ctx, CZK_ERR_HIP, std::string("subgroup check: ") + hipGetErrorString(e));
    *out_bad = h;
    return CZK_OK;
}

template <class F>
static int register_impl(czk_ctx* ctx, czk_bases* b, const u64* pts_dev, const uint8_t* inf_dev) {
    constexpr int AW = GT<F>::AW, JW = GT<F>::JW, FW = GT<F>::FW;
    const size_t n = b->n;
    const unsigned W = b->split ? 1 : b->W;   // windows held as tables
    CZK_HIP(ctx, hipMalloc(&b->pts, (size_t)W * (n ? n : 1) * AW * 8));
#ifdef CZK_LAB   // keys keep saturated tables / the XYZZ kernels on request (A/B runs of the rejected variants)
    const bool keep_sat = ctx->msm_sat || (GT<F>::AW != 12 && ctx->msm_sat_g2), no_te = ctx->msm_sat || ctx->msm_no_te || ctx->msm_affine_rounds > 0;
#else
    constexpr bool keep_sat = false, no_te = false;
#endif
    CZK_HIP(ctx, hipMalloc(&b->inf, (size_t)W * (n ? n : 1)));
    if (!n) {
        b->unsat = !keep_sat;   // an empty key runs the same kernels as any other (its MSMs are the neutral element)
        return CZK_OK;
    }
    CZK_HIP(ctx, hipMemcpyAsync(b->pts, pts_dev, n * AW * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (inf_dev) CZK_HIP(ctx, hipMemcpyAsync(b->inf, inf_dev, n, hipMemcpyDeviceToDevice, ctx->stream));
    else CZK_HIP(ctx, hipMemsetAsync(b->inf, 0, n, ctx->stream));
    if (b->check_wanted) {   // CZK_MEM_CHECK_SUBGROUP: a base outside the prime-order subgroup keeps the handle on the complete XYZZ kernels
        CZK_TRY(subgroup_check_impl<F>(ctx, b->pts, b->inf, n, &b->n_bad));
        b->checked = true;
        if (b->n_bad) b->te_wanted = false;
    }
    if (W > 1) {
        u64 *jac = nullptr, *scr = nullptr;
        CZK_HIP(ctx, hipMalloc(&jac, n * JW * 8));
        CZK_HIP(ctx, hipMalloc(&scr, n * FW * 8));
        const unsigned CH = 32;
        unsigned g1 = (unsigned)((n + 127) / 128), g2 = (unsigned)(((n + CH - 1) / CH + 127) / 128);
        for (unsigned w = 1; w < W; w++) {
            hipLaunchKernelGGL(k_dbl_c<F>, dim3(g1), dim3(128), 0, ctx->stream, b->pts + (size_t)(w - 1) * n * AW, b->inf + (size_t)(w - 1) * n,
                               n, msm_win_width(b->c, msm_full_windows(b->c), w - 1), jac);   // 2^(width of window w - 1) times the previous table
            hipLaunchKernelGGL(k_batch_to_affine<F>, dim3(g2), dim3(128), 0, ctx->stream, jac, n, CH, scr, b->pts + (size_t)w * n * AW,
                               b->inf + (size_t)w * n);
        }
        CZK_HIP(ctx, hipGetLastError());
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        CZK_HIP(ctx, hipFree(jac));
        CZK_HIP(ctx, hipFree(scr));
    }
    if (GT<F>::AW == 12 && b->te_wanted && !no_te) {
        // G1 bases in the prime-order subgroup: window tables as twisted Edwards niels entries (te.h), 7M unified mixed additions
        u64* te = nullptr;
        bool ok = false;
        // The niels table is twice the size of the XYZZ table and is built while that one is still live: a key that fits as XYZZ tables may not fit
        // here.  Running out of memory is not an error -- the handle keeps the XYZZ kernels (same results, ~23 % more arithmetic per addition).
        int rc = te_table_from_sw(ctx, b->pts, b->inf, (size_t)W * n, &te, &ok);
        if (rc == CZK_ERR_NOMEM) {
            (void)hipGetLastError();
            ok = false;
        } else if (rc != CZK_OK) {
            return rc;
        }
        u64* sw0 = nullptr;   // the registered points stay (secondary table sets are built from them)
        if (ok) {
            hipError_t e = hipMalloc(&sw0, n * AW * 8);
            if (e == hipSuccess) e = hipMemcpy(sw0, b->pts, n * AW * 8, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(te);
                if (sw0) (void)hipFree(sw0);
                ok = false;
                if (e != hipErrorOutOfMemory) return set_err(ctx, CZK_ERR_HIP, std::string("registered points: ") + hipGetErrorString(e));
            }
        }
        if (ok) {
            (void)hipFree(b->pts);
            b->pts = te;
            b->pts_sw0 = sw0;
            b->te = true;
            b->unsat = true;
            return CZK_OK;
        }
    }
    if (!keep_sat) {
        // window tables go to the unsaturated residue system of fqu.h (infinity flags are unaffected); the lab build's "msm_sat" /
        // "msm_sat_g2" options keep the saturated kernels for A/B runs
        launch_convert_to_u(ctx->stream, b->pts, (size_t)W * n * (GT<F>::AW / 6));
        CZK_HIP(ctx, hipGetLastError());
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        b->unsat = true;
    }
    return CZK_OK;
}

// ------------------------------------------------------------------------------------------------
// window width per call.  The reference chooses c from the size of each call (variable_base.rs:21-25); with precomputed window
// multiples the width is a property of the table, so a key registered for n points carries c(n) -- and a SHORT MSM under it (a KZG
// commitment of a degree-2^18 polynomial under a 3 * 2^18-point SRS) would still reduce 2^(c(n)-1) buckets per lane: as much
// work as its whole accumulation.  HBM is plentiful, so such calls get their own, narrower table set over the prefix of the
// key they use: classes c = 13 / 15 / 17 by call size (cost model: W(c) size mixed additions + ~6 mixed additions' worth of
// instructions per bucket), covering the next power of two of the call's size, built on first use (or by czk_bases_prepare)
// and kept with the handle.  A set is only built when the model predicts >= 12 % less work than the key's own tables.
// ------------------------------------------------------------------------------------------------
struct TableView {
    unsigned c, W;
    const u64* pts;
    const uint8_t* inf;
    size_t stride;   // points per window
};
constexpr double REDUCE_COST_PER_BUCKET = 6.0;   // in mixed additions (measured: profiles/r03_window_classes.txt)
static double msm_cost(unsigned c, size_t size) { return (double)num_windows(c) * (double)size + REDUCE_COST_PER_BUCKET * (double)((size_t)1 << (c - 1)); }
static unsigned width_class(size_t size) { return size < 11586 ? 13u : size < 92682 ? 15u : 17u; }   // boundaries at 2^13.5, 2^16.5

template <class F>
static int build_secondary(czk_ctx* ctx, const czk_bases* b, unsigned c, size_t cover, czk_table_set* out) {
    constexpr int AW = GT<F>::AW, JW = GT<F>::JW, FW = GT<F>::FW;
    const unsigned W = num_windows(c);
    czk_table_set t;
    t.c = c;
    t.W = W;
    t.cover = cover;
    u64 *jac = nullptr, *scr = nullptr;
    hipError_t e = hipMalloc(&t.pts, (size_t)W * cover * AW * 8);
    if (e == hipSuccess) e = hipMalloc(&t.inf, (size_t)W * cover);
    if (e == hipSuccess) e = hipMalloc(&jac, cover * JW * 8);
    if (e == hipSuccess) e = hipMalloc(&scr, cover * FW * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(t.pts, b->te ? b->pts_sw0 : b->pts, cover * AW * 8, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t.inf, b->inf, cover, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) {
        if (b->unsat && !b->te) launch_convert_from_u(ctx->stream, t.pts, cover * (AW / 6));   // the key's window 0 back to Montgomery form
        const unsigned CH = 32;
        unsigned g1 = (unsigned)((cover + 127) / 128), g2 = (unsigned)(((cover + CH - 1) / CH + 127) / 128);
        for (unsigned w = 1; w < W; w++) {
            hipLaunchKernelGGL(k_dbl_c<F>, dim3(g1), dim3(128), 0, ctx->stream, t.pts + (size_t)(w - 1) * cover * AW, t.inf + (size_t)(w - 1) * cover, cover, msm_win_width(c, msm_full_windows(c), w - 1), jac);
            hipLaunchKernelGGL(k_batch_to_affine<F>, dim3(g2), dim3(128), 0, ctx->stream, jac, cover, CH, scr, t.pts + (size_t)w * cover * AW,
                               t.inf + (size_t)w * cover);
        }
        if (b->unsat && !b->te) launch_convert_to_u(ctx->stream, t.pts, (size_t)W * cover * (AW / 6));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (jac) (void)hipFree(jac);
    if (scr) (void)hipFree(scr);
    if (e == hipSuccess && b->te) {   // the key is in twisted Edwards form: so is this set (its points have images: they are the key's)
        u64* te = nullptr;
        bool ok = false;
        int rc = te_table_from_sw(ctx, t.pts, t.inf, (size_t)W * cover, &te, &ok);
        (void)hipFree(t.pts);
        t.pts = te;
        if (rc != CZK_OK || !ok) {
            if (t.inf) (void)hipFree(t.inf);
            if (te) (void)hipFree(te);
            return rc != CZK_OK ? rc : set_err(ctx, CZK_ERR_ARG, "secondary window tables: a multiple of a registered point has no twisted Edwards image");
        }
    }
    if (e != hipSuccess) {
        if (t.pts) (void)hipFree(t.pts);
        if (t.inf) (void)hipFree(t.inf);
        return set_err(ctx, e == hipErrorOutOfMemory ? CZK_ERR_NOMEM : CZK_ERR_HIP, std::string("secondary window tables: ") + hipGetErrorString(e));
    }
    *out = t;
    return CZK_OK;
}

// the table set an MSM of `size` pairs runs on (builds a secondary set when the model asks for one and none fits)
template <class F>
static int pick_tables(czk_ctx* ctx, const czk_bases* cb, size_t size, TableView* tv, bool build) {
    czk_bases* b = const_cast<czk_bases*>(cb);   // the secondary sets are a cache behind the const handle
    *tv = TableView{b->c, b->W, b->pts, b->inf, b->n};
    if (b->split || !b->per_call_width || size == 0) return CZK_OK;
    const unsigned cc = width_class(size);
    if (cc >= b->c || msm_cost(b->c, size) < 1.12 * msm_cost(cc, size)) return CZK_OK;
    auto find = [&]() -> const czk_table_set* {
        const int n = b->n_extra.load(std::memory_order_acquire);
        const czk_table_set* best = nullptr;
        for (int i = 0; i < n; i++)
            if (b->extra[i].c == cc && b->extra[i].cover >= size && (!best || b->extra[i].cover < best->cover)) best = &b->extra[i];
        return best;
    };
    const czk_table_set* t = find();
    if (!t && build && b->nomem_class.load(std::memory_order_acquire) & (1u << (cc & 31))) return CZK_OK;   // see below: no room last time
    if (!t && build) {
        std::lock_guard<std::mutex> lk(b->build_mu);
        t = find();   // another context may have built it meanwhile
        const int n = b->n_extra.load(std::memory_order_acquire);
        if (!t && n < czk_bases::MAX_EXTRA) {
            // ONE set per narrow class (c = 13: 2^14 points, c = 15: 2^17 -- they are small), powers of two for c = 17: at most 2 + (log2 n - 16) sets per
            // key, so a prover that commits polynomials of many different lengths cannot run out of slots and fall back to the wide tables
            size_t cover = cc == 13 ? ((size_t)1 << 14) : cc == 15 ? ((size_t)1 << 17) : 1;
            while (cover < size) cover <<= 1;
            if (cover > b->n) cover = b->n;
            CZK_TRY(msm_pipeline_sync(ctx));   // (the build synchronises ctx->stream; drain the MSM streams too so that timing stays attributable)
            int rc = build_secondary<F>(ctx, b, cc, cover, &b->extra[n]);
            if (rc == CZK_ERR_NOMEM) {
                // no room for another table set: this call and later ones of its width class run on the key's own tables.  The class is
                // remembered on the handle, so later calls do not drain the pipeline and retry four failing allocations each time
                // (czk_bases_prepare tries again: a caller that freed memory asks for the set explicitly); the swallowed error is cleared.
                (void)hipGetLastError();
                b->nomem_class.fetch_or(1u << (cc & 31), std::memory_order_release);
                ctx->err.clear();
                return CZK_OK;
            }
            CZK_TRY(rc);
            b->n_extra.store(n + 1, std::memory_order_release);
            t = &b->extra[n];
        }
    }
    if (t) *tv = TableView{t->c, t->W, t->pts, t->inf, t->cover};
    return CZK_OK;
}

// two keys of one length and window layout drop the same digits: the same table entries (window w, point i) are infinity
static bool same_infinities(const czk_bases* x, const czk_bases* y) {
    if (x == y) return true;
    return x->inf_listed && y->inf_listed && x->n == y->n && x->c == y->c && x->W == y->W && x->inf_idx == y->inf_idx;
}

// Enqueue one MSM on the context's three-stage pipeline:
//   s_sort : digits -> partition -> per-partition sort -> population order; flag clearing  (memory bound)
//   s_acc  : the bucket accumulation kernel, nothing else                            (integer-VALU bound, fills the chip)
//   s_red  : over-full-bucket items, fix-up kernels, bucket reduction, result copy   (mostly latency bound)
// Consecutive MSMs overlap stage-wise (sort of k+1 and reduce of k-1 hide under accumulate of k); a ring of
// workspace slots is guarded by events.  Results land in a pinned staging area and are handed to the caller's
// buffer by msm_collect() after the streams are synchronised.
template <class F>
static int msm_enqueue(czk_ctx* ctx, const czk_bases* b, const u64* scalars, size_t n_scalars, size_t lanes, int form, u64* out_host,
                       bool scalars_stable, bool reserve_only = false, bool same_scalars = false) {
    constexpr int JW = GT<F>::JW, XW = GT<F>::XW;   // results leave as Jacobian; buckets are XYZZ internally
    const size_t size = b->n < n_scalars ? b->n : n_scalars;   // variable_base.rs:16
    // Bases without window tables (b->split): the W digit windows of every scalar lane become W "virtual lanes", each a
    // one-window MSM over the same n points with its own bucket set; everything after the digit extraction simply runs with
    // lanes * W lanes and one window, and the per-window results are combined on the host when they are collected.
    TableView tv;
    CZK_TRY(pick_tables<F>(ctx, b, size, &tv, true));
    unsigned c = tv.c, Wd = tv.W;
    if (b->split) {   // no tables to honour: the width follows the call, as in the reference (variable_base.rs:21-25)
        c = choose_c_split(size);
        Wd = num_windows(c);
    }
    const unsigned W = b->split ? 1 : Wd;
    const size_t nb = tv.stride;   // points per window of the table in use
    const size_t real_lanes = lanes;
    if (b->split) lanes *= Wd;
    if (lanes > 65535) return set_err(ctx, CZK_ERR_SIZE, "too many MSM lanes");
    const size_t B = (size_t)1 << (c - 1);
    const unsigned L = 8, logL = 3;
    if ((size_t)W * nb >= ((size_t)1 << 31)) return set_err(ctx, CZK_ERR_SIZE, "W * n_bases exceeds the 31-bit point index");
    CZK_TRY(msm_pipeline_init(ctx));
#ifdef CZK_LAB
    if (ctx->chaos && !reserve_only) ctx->msm_next_slot = (int)(chaos_rand(ctx) % (unsigned)ctx->msm_slots_in_use);   // any slot is valid: its events order the reuse
#endif
    const int slot_idx = ctx->msm_next_slot;
    MsmSlot& slot = ctx->msm_slots[slot_idx];
    if (!reserve_only) ctx->msm_next_slot = (ctx->msm_next_slot + 1) % ctx->msm_slots_in_use;

    // workspaces (grow-only; growing synchronises the pipeline first)
    size_t lvl0 = (B + L - 1) / L;
    unsigned n_parts = (unsigned)((B + PART_BUCKETS - 1) >> PART_LOG);   // B is a power of two
    // k_part_sort stages a partition's placement in LDS when it fits (~36 k entries on gfx950): long calls get more, smaller partitions (a partition is
    // the set of buckets with equal LOW index bits, so any power of two up to MAX_PARTS works; the 2^21-point h query: 1024 partitions of 27 k entries)
    bool grow_parts = !b->split;
#ifdef CZK_LAB
    grow_parts = grow_parts && ctx->msm_affine_rounds == 0;   // the batched-affine record builders assume 1024-bucket partitions
#endif
    if (grow_parts)
        while (n_parts < MAX_PARTS && n_parts < B && ((size_t)W * size) / n_parts > 30000) n_parts <<= 1;
    unsigned part_shift = 0;
    while ((1u << part_shift) < n_parts) part_shift++;
    size_t need_sort = lanes * ((size_t)W * size * (4 * 3 + 2) + B * 4 * 4 + CNT_BINS * 4 + (size_t)n_parts * 12 + 64) + (1 << 16);
    const u32 heavy_cap = (u32)(lanes * (size_t)W * size / 256 + 64);   // >= number of 256-entry work items of over-full buckets (msm_acc.h HEAVY_SUB)
    size_t need_red = lanes * (B * XW * 8 + 4 * lvl0 * XW * 8 + JW * 8 + B + (size_t)12 * 513 * XW * 8) + (size_t)heavy_cap * (XW * 8 + 32) + (1 << 17);
    // batched-affine pre-reduction (G1, unsaturated tables): records, two level arrays, per-level bucket offsets / counts
    AffArgs aff;
    const bool one_pass_sort = !b->split && (ctx->msm_sort_onepass || n_parts > MAX_PARTS);   // (choose_c_split keeps Wd * n_parts <= MAX_PARTS)
    size_t need_aff = 0;
    // G1 buckets stay in the unsaturated residue system through the reduction (k_reduce_*_u) unless the batched-affine rounds
    // (which finish on saturated level points) or CZK_REDUCE_SAT ask for the saturated form
    const int te = b->te ? 1 : 0;   // twisted Edwards tables and buckets (te.h): unified additions, no exception handling at all
#ifdef CZK_LAB
    const int ub = (te || (b->unsat && !(GT<F>::AW == 12 ? ctx->msm_reduce_sat || (ctx->msm_affine_rounds > 0 && size > 0) : ctx->msm_reduce_sat || ctx->msm_reduce_sat_g2))) ? 1 : 0;
#else
    const int ub = 1;   // product build: every key's tables are in the unsaturated residue system, buckets stay in u-form through the reduction
#endif
#ifdef CZK_LAB
    if (GT<F>::AW == 12 && b->unsat && !te && ctx->msm_affine_rounds > 0 && size > 0) {
        aff.rounds = ctx->msm_affine_rounds;
        aff.lanes = (unsigned)lanes;
        aff.B = B;
        aff.sorted_stride = (size_t)W * size;
        aff.n_parts = one_pass_sort ? 0 : n_parts;
        aff.part_shift = part_shift;
        aff.part_log = PART_LOG;
        aff_plan((size_t)W * size, B, aff.rounds, aff.S);
        for (unsigned r = 0; r < aff.rounds; r++) need_aff += lanes * aff.S[r] * (8 + 1) + 512;
        need_aff += lanes * aff.S[0] * 128 + (aff.rounds > 1 ? lanes * aff.S[1] * 128 : 0) + 4 * lanes * B * 4 + aff_scratch_bytes(ctx) + (1 << 16);
    }
#endif
    if (slot.ws_sort.bytes < need_sort || slot.ws_red.bytes < need_red || slot.ws_aff.bytes < need_aff) {
        // growing drains the pipeline and calls hipMalloc (a device-wide synchronisation): grow EVERY slot of the ring to the new size at once,
        // so that a prover whose MSMs differ in size (KZG commitments of many lengths) stalls once per new maximum, not once per slot
        CZK_TRY(msm_pipeline_sync(ctx));
        for (int i = 0; i < ctx->msm_slots_in_use; i++) {
            MsmSlot& sl = ctx->msm_slots[i];
            CZK_TRY(ensure_buf(ctx, sl.ws_sort, need_sort));
            CZK_TRY(ensure_buf(ctx, sl.ws_red, need_red));
            if (need_aff) CZK_TRY(ensure_buf(ctx, sl.ws_aff, need_aff));
        }
    }
    if (reserve_only) return CZK_OK;   // czk_ctx_reserve: table set chosen (and built), streams created, every workspace of the ring sized
#ifdef CZK_LAB
    if (aff.rounds) {
        Bump ba{(char*)slot.ws_aff.p};
        for (unsigned r = 0; r < aff.rounds; r++) {
            aff.rec[r] = ba.take<u64>(lanes * aff.S[r]);
            aff.pend[r] = ba.take<uint8_t>(lanes * aff.S[r]);
        }
        aff.lvl[0] = ba.take<char>(lanes * aff.S[0] * 128);
        aff.lvl[1] = aff.rounds > 1 ? ba.take<char>(lanes * aff.S[1] * 128) : nullptr;
        for (int k = 0; k < 2; k++) {
            aff.off[k] = ba.take<u32>(lanes * B);
            aff.cnt[k] = ba.take<u32>(lanes * B);
        }
        aff.scratch = ba.take<char>(aff_scratch_bytes(ctx));
    }
#endif
    // CZK_MEM_SAME_SCALARS: an earlier call's digit sort stands for this one when nothing that enters it differs -- scalars, their form, the window layout, the
    // table stride (entries are table indices) and the digits dropped for points at infinity.  The entries stay in THAT call's slot (`src`); this call takes its own
    // slot for the buckets and the reduction as always.
    const uint64_t seq = ++ctx->msm_seq;
    if (!same_scalars) ctx->msm_leader_seq = seq;
    int src_idx = -1;
    if (same_scalars && scalars_stable && ctx->msm_sort_reuse && !b->split && tv.pts == b->pts) {   // (the primary table set)
        for (int i = 0; i < ctx->msm_slots_in_use && src_idx < 0; i++) {
            const MsmSortKey& k = ctx->msm_slots[i].key;
            bool ok = k.valid && k.seq >= ctx->msm_leader_seq && k.scalars == (const void*)scalars && k.n_scalars == n_scalars && k.lanes == lanes && k.size == size && k.nb == nb && k.form == form &&
                      k.c == c && k.W == W;
#ifdef CZK_LAB
            if (aff.rounds || ctx->chaos_drop_wait) ok = false;
            if (ok && !ctx->msm_sort_reuse_any_inf) ok = same_infinities(k.bases, b);
#else
            if (ok) ok = same_infinities(k.bases, b);
#endif
            if (ok) src_idx = i;
        }
    }
    const bool reuse = src_idx >= 0;
    MsmSlot& src = reuse ? ctx->msm_slots[src_idx] : slot;
    if (!reuse) {
        MsmSortKey& k = slot.key;
        k.valid = scalars_stable && !b->split && tv.pts == b->pts;
        k.seq = seq;
        k.scalars = scalars;
        k.n_scalars = n_scalars;
        k.lanes = lanes;
        k.size = size;
        k.nb = nb;
        k.form = form;
        k.c = c;
        k.W = W;
        k.bases = b;
    }
    Bump bs{(char*)src.ws_sort.p};
    u32* digits = bs.take<u32>(lanes * W * size);
    u32* sorted = bs.take<u32>(lanes * W * size);
    u32* ranks = bs.take<u32>(lanes * W * size);            // one-pass sort: ranks; partitioned sort: entries grouped by partition
    uint16_t* part_lb = bs.take<uint16_t>(lanes * W * size);
    u32* part_counts = bs.take<u32>(lanes * n_parts);
    u32* part_cursor = bs.take<u32>(lanes * n_parts);
    u32* part_base = bs.take<u32>(lanes * (n_parts + 1));
    u32* counts = bs.take<u32>(lanes * B);
    u32* offsets = bs.take<u32>(lanes * B);
    u32* perm = bs.take<u32>(lanes * B);
    u32* chist = bs.take<u32>(lanes * CNT_BINS);
    const size_t n_tiles = (B + SCAN_TILE - 1) / SCAN_TILE;
    u32* tile_sums = bs.take<u32>(lanes * n_tiles);
    Bump br{(char*)slot.ws_red.p};
    u64* buckets = br.take<u64>(lanes * B * XW);
    u64* lv[4];
    for (int i = 0; i < 4; i++) lv[i] = br.take<u64>(lanes * lvl0 * XW);
    u64* result = br.take<u64>(lanes * JW);
    u64* tail_scratch = br.take<u64>(lanes * 12 * 512 * XW);   // reduction tail (G1): tree-reduction scratch and the 12 sums
    u64* tail_sums = br.take<u64>(lanes * 12 * XW);
    u64* heavy_partials = br.take<u64>((size_t)heavy_cap * XW);   // over-full buckets: per-chunk partial sums, item / bucket lists, header
    u32* heavy_items = br.take<u32>((size_t)heavy_cap * 3);
    u32* heavy_list = br.take<u32>((size_t)heavy_cap * 4);
    u32* heavy_hdr = br.take<u32>(4);
    uint8_t* dirty = br.take<uint8_t>(lanes * B + 64 + 3 * 4096 * 4 + 64);   // unsaturated kernel: dirty flags + exception list

    // pinned staging for the result
    const size_t out_bytes = lanes * JW * 8;
    char* pinned = nullptr;
    CZK_TRY(msm_pinned_take(ctx, out_bytes, &pinned));

    hipStream_t ss = ctx->s_sort, sa = ctx->s_acc, sr = ctx->s_red;
    // inputs (scalars) are produced on the caller's stream
    CZK_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->stream));
    CZK_HIP(ctx, hipStreamWaitEvent(ss, ctx->ev_in, 0));
    if (slot.used) CZK_HIP(ctx, hipStreamWaitEvent(ss, slot.ev_fix, 0));   // slot's sort buffers are read by its accumulate and fix-up kernels
#ifdef CZK_LAB
    if (ctx->chaos_drop_wait && slot.used) CZK_HIP(ctx, hipStreamWaitEvent(ss, slot.ev_acc, 0));   // (the broken schedule below breaks bucket CONTENTS only: ev_fix no longer implies ev_acc there)
#endif
    {
        ProfScope ps(ctx, reuse ? "msm_sort_reused" : "msm_sort", ss);
        const bool one_pass = one_pass_sort;
        if (reuse) {   // (`ss` is in order: the entries of `src` are complete before anything recorded on it from here)
        } else if (one_pass) {
            CZK_HIP(ctx, hipMemsetAsync(counts, 0, lanes * B * 4, ss));
            if (size) {
                hipLaunchKernelGGL(k_digits, dim3((unsigned)((size + 255) / 256), (unsigned)lanes), dim3(256), 0, ss, scalars, n_scalars, size,
                                   form == CZK_SCALAR_MONTGOMERY ? 1 : 0, c, W, tv.inf, nb, digits, ranks, counts, B);
            }
            hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles, (unsigned)lanes), dim3(256), 0, ss, counts, B, tile_sums, n_tiles);
            hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)lanes), dim3(1024), 0, ss, tile_sums, n_tiles);
            hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles, (unsigned)lanes), dim3(256), 0, ss, counts, offsets, B, tile_sums, n_tiles);
            if (size) {
                hipLaunchKernelGGL(k_scatter, dim3((unsigned)(((size_t)W * size + 255) / 256), (unsigned)lanes), dim3(256), 0, ss, digits, ranks, size, W,
                                   nb, offsets, B, sorted);
            }
        } else {
            const size_t total = (size_t)W * size;
            CZK_HIP(ctx, hipMemsetAsync(part_counts, 0, lanes * n_parts * 4, ss));
            if (size) {
                hipLaunchKernelGGL(k_digits_part, dim3((unsigned)((size + 255) / 256), (unsigned)real_lanes), dim3(256), 0, ss, scalars, n_scalars, size,
                                   form == CZK_SCALAR_MONTGOMERY ? 1 : 0, c, Wd, tv.inf, nb, digits, part_counts, n_parts, b->split ? 1 : 0);
            }
            hipLaunchKernelGGL(k_part_scan, dim3((unsigned)lanes), dim3(1024), 0, ss, part_counts, part_base, part_cursor, n_parts);
            if (size) {
                if (n_parts >= 2048 && part_scatter_lds(1024, n_parts) <= ctx->lds_per_block)
                    hipLaunchKernelGGL(k_part_scatter<1024>, dim3((unsigned)((total + 1024 * PS_TILE - 1) / (1024 * PS_TILE)), (unsigned)lanes), dim3(1024),
                                       part_scatter_lds(1024, n_parts), ss, digits, size, W, nb, part_base, part_cursor, n_parts, part_shift, ranks, part_lb);
                else if ((n_parts == 1024 || n_parts == 512) && part_scatter_lds(512, n_parts) <= ctx->lds_per_block)
                    hipLaunchKernelGGL(k_part_scatter<512>, dim3((unsigned)((total + 512 * PS_TILE - 1) / (512 * PS_TILE)), (unsigned)lanes), dim3(512),
                                       part_scatter_lds(512, n_parts), ss, digits, size, W, nb, part_base, part_cursor, n_parts, part_shift, ranks, part_lb);
                else
                    hipLaunchKernelGGL(k_part_scatter<256>, dim3((unsigned)((total + 256 * PS_TILE - 1) / (256 * PS_TILE)), (unsigned)lanes), dim3(256),
                                       part_scatter_lds(256, n_parts), ss, digits, size, W, nb, part_base, part_cursor, n_parts, part_shift, ranks, part_lb);
            }
            {
                // dynamic LDS: counters + scan scratch + a staging area -- as large as the device's per-workgroup limit allows (gfx950: 160 KiB -> 36 k
                // entries) for long calls, but no larger than a partition can need (twice the mean partition + slack; a partition beyond the area
                // places directly): a 3-point commitment or a 2^13-point call has a few hundred entries per partition, and a 160 KiB request would
                // pin one workgroup per CU and block LDS for the kernels of the other contexts on the GPU
                const size_t fixed = (2 * PART_BUCKETS + PSORT_THREADS) * 4;
                const size_t limit = ctx->lds_per_block > fixed + 4096 ? ctx->lds_per_block : fixed + 4096;
                size_t want = 2 * (total / n_parts) + 1024;
                if (want > total) want = total;
                size_t lds = fixed + (want * 4 > 4096 ? want * 4 : 4096);
                if (lds > limit) lds = limit;
                const u32 cap = (u32)((lds - fixed) / 4);
                hipLaunchKernelGGL(k_part_sort, dim3(n_parts, (unsigned)lanes), dim3(PSORT_THREADS), lds, ss, ranks, part_lb, part_base, n_parts, part_shift, total, B,
                                   sorted, offsets, counts, cap);
            }
        }
        if (!reuse) {
            CZK_HIP(ctx, hipMemsetAsync(chist, 0, lanes * CNT_BINS * 4, ss));
            hipLaunchKernelGGL(k_count_hist, dim3((unsigned)((B + 1023) / 1024), (unsigned)lanes), dim3(1024), 0, ss, counts, B, chist);
            hipLaunchKernelGGL(k_count_starts, dim3((unsigned)lanes), dim3(1024), 0, ss, chist);
            hipLaunchKernelGGL(k_count_scatter, dim3((unsigned)((B + 1023) / 1024), (unsigned)lanes), dim3(1024), 0, ss, counts, B, chist, perm);
        }
#ifdef CZK_LAB
        if (aff.rounds) {
            aff.sorted = sorted;
            aff.offsets = offsets;
            aff.counts = counts;
            launch_affine_build_g1(ss, aff);
        }
#endif
        if (b->unsat && !te) {
            // clear the dirty flags / exception list here rather than on the accumulate stream (the critical one); they live
            // in the slot's reduce workspace, which the slot's previous reduction may still be using
            if (slot.used) CZK_HIP(ctx, hipStreamWaitEvent(ss, slot.ev_red, 0));
            if (GT<F>::AW == 12) launch_accumulate_g1_u_prepare(ss, dirty, B, (unsigned)lanes);
            else launch_accumulate_g2_u_prepare(ss, dirty, B, (unsigned)lanes);
        }
    }
    CZK_HIP(ctx, hipEventRecord(slot.ev_sorted, ss));
    // the caller's stream may overwrite the scalars once the digits are extracted
    if (!scalars_stable) CZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, slot.ev_sorted, 0));
    CZK_HIP(ctx, hipStreamWaitEvent(sa, slot.ev_sorted, 0));
    if (slot.used) CZK_HIP(ctx, hipStreamWaitEvent(sa, slot.ev_red, 0));   // slot's buckets are read by its reduce
    ctx->msm_launch_split = b->split;
    if (te) {
        launch_accumulate_g1_te(ctx, sa, tv.pts, sorted, offsets, counts, perm, B, (size_t)W * size, buckets, (unsigned)lanes);
    } else if (b->unsat) {
        // (these launchers bracket their main kernel with the "msm_accumulate_g{1,2}" profiling scope themselves)
#ifdef CZK_LAB
        if (aff.rounds) launch_affine_accumulate_g1(ctx, sa, aff, tv.pts, perm, buckets, dirty);
        else
#endif
        if (GT<F>::AW == 12) launch_accumulate_g1_u(ctx, sa, tv.pts, sorted, offsets, counts, perm, B, (size_t)W * size, buckets, (unsigned)lanes, dirty, ub);
        else launch_accumulate_g2_u(ctx, sa, tv.pts, sorted, offsets, counts, perm, B, (size_t)W * size, buckets, (unsigned)lanes, dirty, ub);
    } else {
#ifdef CZK_LAB   // saturated tables ("msm_sat"): the round-1 kernels
        ProfScope ps(ctx, GT<F>::AW == 12 ? "msm_accumulate_g1" : "msm_accumulate_g2", sa);
        if (GT<F>::AW == 12) launch_accumulate_g1(sa, tv.pts, sorted, offsets, counts, perm, B, (size_t)W * size, buckets, (unsigned)lanes);
        else launch_accumulate_g2(sa, tv.pts, sorted, offsets, counts, perm, B, (size_t)W * size, buckets, (unsigned)lanes);
#else
        return set_err(ctx, CZK_ERR_ARG, "bases with saturated window tables: not part of the product build");
#endif
    }
    CZK_HIP(ctx, hipGetLastError());
    CZK_HIP(ctx, hipEventRecord(slot.ev_acc, sa));
#ifdef CZK_LAB
    // the deliberately broken schedule of tests/test_chaos.py: the reduce stream waits for the digit sort only, not for the accumulate kernel -- it then folds buckets
    // that are stale or half written (wrong results, which the harness must notice), while everything else it reads (entry lists, counts, offsets) is complete and
    // well formed, and the slot's sort buffers stay protected (the extra wait for ev_acc above).  (Dropping the accumulate stream's wait for the sort instead lets
    // kernels read another call's buffer layout as counts and indices: memory faults and minute-long loops -- that was the first version of this switch.)
    if (ctx->chaos_drop_wait) CZK_HIP(ctx, hipStreamWaitEvent(sr, slot.ev_sorted, 0));
    else
#endif
    CZK_HIP(ctx, hipStreamWaitEvent(sr, slot.ev_acc, 0));
    // work items beyond the first 1024 entries of over-full buckets (none with uniformly random scalars): reduce stream
    if (te) launch_heavy_g1_te(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, heavy_hdr, heavy_items, heavy_list, heavy_partials,
                               heavy_cap);
    else if (GT<F>::AW == 12) launch_heavy_g1(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, b->unsat ? dirty : nullptr, heavy_hdr,
                                         heavy_items, heavy_list, heavy_partials, heavy_cap, b->unsat ? 1 : 0, ub);
    else launch_heavy_g2(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, b->unsat ? dirty : nullptr, heavy_hdr, heavy_items,
                         heavy_list, heavy_partials, heavy_cap, b->unsat ? 1 : 0, ub);
    if (b->unsat && !te) {
        // dirty buckets / deferred points (normally none): on the reduce stream, so the accumulate stream goes straight on
#ifdef CZK_LAB
        if (aff.rounds)
            launch_accumulate_g1_u_fixup_lvl(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, dirty, aff.lvl[(aff.rounds - 1) & 1]);
        else
#endif
        if (GT<F>::AW == 12) launch_accumulate_g1_u_fixup(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, dirty, ub);
        else launch_accumulate_g2_u_fixup(sr, tv.pts, sorted, offsets, counts, B, (size_t)W * size, buckets, (unsigned)lanes, dirty, ub);
    }
    CZK_HIP(ctx, hipEventRecord(slot.ev_fix, sr));   // the slot's sort buffers are free from here
    if (reuse && &src != &slot) CZK_HIP(ctx, hipEventRecord(src.ev_fix, sr));   // ... and so are the ones this call borrowed: the next sort into `src` waits for the
                                                                                 // LATEST record (`sr` is in order, so it covers src's own accumulate and fix-up too)
    {
        ProfScope ps(ctx, "msm_reduce", sr);
        const u64 *P = buckets, *E = nullptr;
        size_t n_in = B;
        unsigned level = 0;
        int flip = 0;
        bool finished = false;
        while (n_in > 1) {
            if (n_in <= 1024) {   // latency-bound from here: bit-sum tree reductions instead of more levels
                if (GT<F>::AW != 12) launch_reduce_tail_g2(sr, P, E, n_in, level * logL, tail_scratch, tail_sums, result, (unsigned)lanes, ub);
                else if (ub) launch_reduce_tail_g1_u(sr, P, E, n_in, level * logL, tail_scratch, tail_sums, result, (unsigned)lanes, te);
#ifdef CZK_LAB
                else launch_reduce_tail_g1(sr, P, E, n_in, level * logL, tail_scratch, tail_sums, result, (unsigned)lanes);
#endif
                finished = true;
                break;
            }
            size_t n_out = (n_in + L - 1) / L;
            u64 *Po = lv[flip * 2], *Eo = lv[flip * 2 + 1];
            if (GT<F>::AW != 12) launch_reduce_level_g2(sr, P, E, n_in, L, level * logL, Po, Eo, n_out, (unsigned)lanes, ub);
            else if (ub) launch_reduce_level_g1_u(sr, P, E, n_in, L, level * logL, Po, Eo, n_out, (unsigned)lanes, te);
#ifdef CZK_LAB
            else launch_reduce_level_g1(sr, P, E, n_in, L, level * logL, Po, Eo, n_out, (unsigned)lanes);
#endif
            P = Po;
            E = Eo;
            n_in = n_out;
            level++;
            flip ^= 1;
        }
        if (finished) {
        } else if (GT<F>::AW != 12) launch_finish_g2(sr, P, E, lanes, result, ub);
        else if (ub) launch_finish_g1_u(sr, P, E, lanes, result, te);
#ifdef CZK_LAB
        else launch_finish_g1(sr, P, E, lanes, result);
#endif
    }
    CZK_HIP(ctx, hipGetLastError());
    chaos_point(ctx, sr);
    CZK_HIP(ctx, hipMemcpyAsync(pinned, result, out_bytes, hipMemcpyDeviceToHost, sr));
    CZK_HIP(ctx, hipEventRecord(slot.ev_red, sr));
    slot.used = true;
    MsmPending pend{pinned, out_host, out_bytes};
    if (b->split) {
        pend.split_W = Wd;
        pend.c = c;
        pend.group = b->group;
        pend.lanes = real_lanes;
    }
    ctx->msm_pending.push_back(pend);
    return CZK_OK;
}

int msm_pipeline_init(czk_ctx* ctx) {
    if (ctx->s_sort) return CZK_OK;
    // (stream priorities for the short sort / reduce stages were measured: no gain, so all three are equal)
    if (ctx->msm_stream_prio) {   // option "msm_stream_priority": 1 = sort / reduce streams above the accumulate stream, 2 = the reverse
        int least = 0, greatest = 0;
        CZK_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        const bool rev = ctx->msm_stream_prio == 2;
        CZK_HIP(ctx, hipStreamCreateWithPriority(&ctx->s_sort, hipStreamNonBlocking, rev ? least : greatest));
        CZK_HIP(ctx, hipStreamCreateWithPriority(&ctx->s_acc, hipStreamNonBlocking, rev ? greatest : least));
        CZK_HIP(ctx, hipStreamCreateWithPriority(&ctx->s_red, hipStreamNonBlocking, rev ? least : greatest));
    } else {
        CZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_sort, hipStreamNonBlocking));
        CZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_acc, hipStreamNonBlocking));
        CZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_red, hipStreamNonBlocking));
    }
    CZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
    for (auto& s : ctx->msm_slots) {
        CZK_HIP(ctx, hipEventCreateWithFlags(&s.ev_sorted, hipEventDisableTiming));
        CZK_HIP(ctx, hipEventCreateWithFlags(&s.ev_acc, hipEventDisableTiming));
        CZK_HIP(ctx, hipEventCreateWithFlags(&s.ev_fix, hipEventDisableTiming));
        CZK_HIP(ctx, hipEventCreateWithFlags(&s.ev_red, hipEventDisableTiming));
    }
    ctx->msm_pinned_bytes = 4 << 20;
    CZK_HIP(ctx, hipHostMalloc((void**)&ctx->msm_pinned, ctx->msm_pinned_bytes, hipHostMallocDefault));
    return CZK_OK;
}

static void deliver(const MsmPending& p) {
    if (p.split_W) host_combine_windows(p.group, p.src, p.split_W, p.c, p.lanes, (uint64_t*)p.dst);
    else memcpy(p.dst, p.src, p.bytes);
}
static void retire_mark(czk_ctx* ctx, CtxMark& m) {
    ctx->mark_events.push_back(m.ev_stream);
    ctx->mark_events.push_back(m.ev_red);
}

// wait for every enqueued MSM and deliver the results (and those of deferred downloads: their copies run on the context's stream)
int msm_pipeline_sync(czk_ctx* ctx) {
    if (!ctx->s_sort) return CZK_OK;
    for (auto& p : ctx->msm_pending)
        if (p.on_stream) {
            CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            break;
        }
    CZK_HIP(ctx, hipStreamSynchronize(ctx->s_sort));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->s_acc));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->s_red));
    for (auto& p : ctx->msm_pending) deliver(p);
    for (auto& sl : ctx->msm_slots) sl.key.valid = false;   // CZK_MEM_STABLE promises the scalars up to here
    ctx->msm_delivered += ctx->msm_pending.size();
    ctx->msm_pending.clear();
    ctx->msm_pinned_used = 0;
    for (auto& m : ctx->marks) retire_mark(ctx, m);   // everything a mark could name is done
    ctx->marks.clear();
    return CZK_OK;
}

// Staging for one host result: a ring over the pinned area.  Results are delivered oldest first, so the free space runs from the tail (`used`) round to
// the oldest pending result; when a request does not fit, everything pending is drained (a full synchronisation -- 4 MiB hold > 10^4 MSM results).
int msm_pinned_take(czk_ctx* ctx, size_t bytes, char** out) {
    CZK_TRY(msm_pipeline_init(ctx));
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > ctx->msm_pinned_bytes) return set_err(ctx, CZK_ERR_SIZE, "host result larger than the result staging area");
    if (ctx->msm_pending.empty()) ctx->msm_pinned_used = 0;
    else {
        const size_t head = (size_t)(ctx->msm_pending.front().src - ctx->msm_pinned), tail = ctx->msm_pinned_used;
        bool fits;
        if (tail > head) {                         // [head, tail) in use
            fits = tail + need <= ctx->msm_pinned_bytes;
            if (!fits && need <= head) {           // wrap: [0, head) is free
                ctx->msm_pinned_used = 0;
                fits = true;
            }
        } else fits = tail + need <= head;         // wrapped: [tail, head) is free (tail == head with results pending: full)
        if (!fits) {
            CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            CZK_TRY(msm_pipeline_sync(ctx));
        }
    }
    *out = ctx->msm_pinned + ctx->msm_pinned_used;
    ctx->msm_pinned_used += need;
    return CZK_OK;
}

// ---- marks (include/czk.h: czk_ctx_mark / czk_ctx_wait_mark / czk_lanes_download_deferred) -------------------------------------------
static hipEvent_t mark_event(czk_ctx* ctx) {
    if (!ctx->mark_events.empty()) {
        hipEvent_t e = ctx->mark_events.back();
        ctx->mark_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return e;
}
int ctx_mark(czk_ctx* ctx, uint64_t* out) {
    CZK_TRY(msm_pipeline_init(ctx));
    CtxMark m;
    m.id = ctx->next_mark++;
    m.ev_stream = mark_event(ctx);
    m.ev_red = mark_event(ctx);
    if (!m.ev_stream || !m.ev_red) return set_err(ctx, CZK_ERR_HIP, "czk_ctx_mark: no event");
    CZK_HIP(ctx, hipEventRecord(m.ev_stream, ctx->stream));
    CZK_HIP(ctx, hipEventRecord(m.ev_red, ctx->s_red));   // every MSM enqueued so far ends with its result copy on this stream, in order
    m.upto = ctx->msm_delivered + ctx->msm_pending.size();
    ctx->marks.push_back(m);
    *out = m.id;
    return CZK_OK;
}
int ctx_wait_mark(czk_ctx* ctx, uint64_t id) {
    if (id == 0 || id >= ctx->next_mark) return set_err(ctx, CZK_ERR_ARG, "czk_ctx_wait_mark: not a mark of this context");
    for (auto& sl : ctx->msm_slots) sl.key.valid = false;   // a transcript point: what follows is another round's (or another proof's) work
    while (!ctx->marks.empty() && ctx->marks.front().id <= id) {   // older marks first: a mark covers the ones before it
        CtxMark m = ctx->marks.front();
        CZK_HIP(ctx, hipEventSynchronize(m.ev_stream));
        CZK_HIP(ctx, hipEventSynchronize(m.ev_red));
        size_t k = 0;
        while (k < ctx->msm_pending.size() && ctx->msm_delivered + k < m.upto) deliver(ctx->msm_pending[k++]);
        ctx->msm_pending.erase(ctx->msm_pending.begin(), ctx->msm_pending.begin() + k);
        ctx->msm_delivered += k;
        retire_mark(ctx, m);
        ctx->marks.erase(ctx->marks.begin());
    }
    return CZK_OK;   // (a mark retired earlier -- by a later mark's wait or by czk_ctx_sync -- has nothing left to wait for)
}

void msm_pipeline_destroy(czk_ctx* ctx) {
    if (!ctx->s_sort) return;
    // drain the streams, but do NOT deliver pending czk_msm_async results: the caller's `out_jac` buffers may be gone by
    // now (an exception between msm_async and sync, then garbage collection in any order); results are only ever
    // delivered by an explicit czk_ctx_sync / czk_msm
    (void)hipStreamSynchronize(ctx->s_sort);
    (void)hipStreamSynchronize(ctx->s_acc);
    (void)hipStreamSynchronize(ctx->s_red);
    ctx->msm_pending.clear();
    for (auto& m : ctx->marks) retire_mark(ctx, m);
    ctx->marks.clear();
    for (hipEvent_t e : ctx->mark_events) (void)hipEventDestroy(e);
    ctx->mark_events.clear();
    for (auto& s : ctx->msm_slots) {
        if (s.ws_sort.p) (void)hipFree(s.ws_sort.p);
        if (s.ws_red.p) (void)hipFree(s.ws_red.p);
        if (s.ws_aff.p) (void)hipFree(s.ws_aff.p);
        (void)hipEventDestroy(s.ev_sorted);
        (void)hipEventDestroy(s.ev_acc);
        (void)hipEventDestroy(s.ev_fix);
        (void)hipEventDestroy(s.ev_red);
    }
    (void)hipEventDestroy(ctx->ev_in);
    (void)hipHostFree(ctx->msm_pinned);
    (void)hipStreamDestroy(ctx->s_sort);
    (void)hipStreamDestroy(ctx->s_acc);
    (void)hipStreamDestroy(ctx->s_red);
    ctx->s_sort = nullptr;
}

int msm_reserve(czk_ctx* ctx, const czk_bases* bases, size_t n_scalars, size_t lanes) {
    return bases->group == CZK_G1 ? msm_enqueue<Fq>(ctx, bases, nullptr, n_scalars, lanes, CZK_SCALAR_CANONICAL, nullptr, false, true)
                                  : msm_enqueue<Fq2>(ctx, bases, nullptr, n_scalars, lanes, CZK_SCALAR_CANONICAL, nullptr, false, true);
}

int msm_device(czk_ctx* ctx, const czk_bases* bases, const u64* scalars_dev, size_t n_scalars, size_t lanes, int scalar_form,
               u64* out_jac_host, bool blocking, bool scalars_stable, bool same_scalars) {
    int rc = bases->group == CZK_G1 ? msm_enqueue<Fq>(ctx, bases, scalars_dev, n_scalars, lanes, scalar_form, out_jac_host, scalars_stable, false, same_scalars)
                                    : msm_enqueue<Fq2>(ctx, bases, scalars_dev, n_scalars, lanes, scalar_form, out_jac_host, scalars_stable, false, same_scalars);
    if (rc != CZK_OK || !blocking) return rc;
    return msm_pipeline_sync(ctx);
}

template <class F>
static int fixed_base_impl(czk_ctx* ctx, const u64* k_dev, size_t n, u64* out_dev) {
    constexpr int JW = GT<F>::JW, FW = GT<F>::FW;
    if (!n) return CZK_OK;
    u64 *jac = nullptr, *scr = nullptr;
    uint8_t* inf = nullptr;
    CZK_HIP(ctx, hipMalloc(&jac, n * JW * 8));
    CZK_HIP(ctx, hipMalloc(&scr, n * FW * 8));
    CZK_HIP(ctx, hipMalloc(&inf, n));
    const unsigned CH = 32;
    hipLaunchKernelGGL(k_fixed_base<F>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, k_dev, n, jac);
    hipLaunchKernelGGL(k_batch_to_affine<F>, dim3((unsigned)(((n + CH - 1) / CH + 127) / 128)), dim3(128), 0, ctx->stream, jac, n, CH, scr,
                       out_dev, inf);
    CZK_HIP(ctx, hipGetLastError());
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CZK_HIP(ctx, hipFree(jac));
    CZK_HIP(ctx, hipFree(scr));
    CZK_HIP(ctx, hipFree(inf));
    return CZK_OK;
}

int fixed_base_points_device(czk_ctx* ctx, int group, const u64* k_dev, size_t n, u64* out_dev) {
    return group == CZK_G1 ? fixed_base_impl<Fq>(ctx, k_dev, n, out_dev) : fixed_base_impl<Fq2>(ctx, k_dev, n, out_dev);
}

}  // namespace czk

using namespace czk;

// ------------------------------------------------------------------------------------------------
// C ABI (MSM part)
// ------------------------------------------------------------------------------------------------
extern "C" int czk_bases_register(czk_ctx* ctx, int group, const uint64_t* bases, const uint8_t* inf, size_t n, int mem, czk_bases** out) {
    if (!ctx || !out) return CZK_ERR_ARG;
    *out = nullptr;
    if (group != CZK_G1 && group != CZK_G2) return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    if (n && !bases) return set_err(ctx, CZK_ERR_ARG, "null bases");
    const bool no_tables = (mem & CZK_MEM_NO_TABLES) != 0, any_points = (mem & CZK_MEM_ANY_POINTS) != 0, check = (mem & CZK_MEM_CHECK_SUBGROUP) != 0;
    mem &= ~(CZK_MEM_NO_TABLES | CZK_MEM_ANY_POINTS | CZK_MEM_CHECK_SUBGROUP);
    if (!valid_mem(mem))
        return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE (optionally | CZK_MEM_NO_TABLES | CZK_MEM_ANY_POINTS | CZK_MEM_CHECK_SUBGROUP)");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t aw = group == CZK_G1 ? 12 : 24;
    czk_bases* b = new czk_bases();
    b->device = ctx->device;
    b->group = group;
    b->n = n;
    b->split = no_tables;
    b->te_wanted = !any_points;
    b->check_wanted = check;
    b->per_call_width = !ctx->msm_fixed_c;
    b->c = no_tables ? choose_c_split(n) : choose_c(n);
    if (const unsigned v = group == CZK_G1 ? ctx->msm_c_g1 : ctx->msm_c_g2)   // option "msm_window_g1" / "_g2": the primary table set's width
        if (!no_tables && v >= 8 && v <= 22) b->c = v;
    b->W = num_windows(b->c);
    const u64* pts_dev = bases;
    const uint8_t* inf_dev = inf;
    void *tmp_p = nullptr, *tmp_i = nullptr;
    int rc = CZK_OK;
    if (mem == CZK_MEM_HOST && n) {
        if (hipMalloc(&tmp_p, n * aw * 8) != hipSuccess) rc = set_err(ctx, CZK_ERR_NOMEM, "hipMalloc bases staging");
        if (rc == CZK_OK && hipMemcpyAsync(tmp_p, bases, n * aw * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = set_err(ctx, CZK_ERR_HIP, "H2D bases");
        pts_dev = (const u64*)tmp_p;
        if (rc == CZK_OK && inf) {
            if (hipMalloc(&tmp_i, n) != hipSuccess) rc = set_err(ctx, CZK_ERR_NOMEM, "hipMalloc inf staging");
            if (rc == CZK_OK && hipMemcpyAsync(tmp_i, inf, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = set_err(ctx, CZK_ERR_HIP, "H2D inf");
            inf_dev = (const uint8_t*)tmp_i;
        }
    }
    if (rc == CZK_OK) rc = group == CZK_G1 ? register_impl<Fq>(ctx, b, pts_dev, inf_dev) : register_impl<Fq2>(ctx, b, pts_dev, inf_dev);
    (void)hipStreamSynchronize(ctx->stream);
    if (tmp_p) (void)hipFree(tmp_p);
    if (tmp_i) (void)hipFree(tmp_i);
    const size_t n_flags = (no_tables ? 1 : (size_t)b->W) * n;   // (a multiple 2^(c w) P of a point outside the subgroup can be infinity where P is not: every window counts)
    if (rc == CZK_OK && !ctx->msm_sort_reuse) {   // (the lists are only read under the option: a key registered without it never shares a sort)
    } else if (rc == CZK_OK && n && n_flags < ((size_t)1 << 32)) {   // which table entries are infinity (CZK_MEM_SAME_SCALARS compares two keys' lists)
        std::vector<uint8_t> flags(n_flags);
        if (hipMemcpy(flags.data(), b->inf, n_flags, hipMemcpyDeviceToHost) != hipSuccess) rc = set_err(ctx, CZK_ERR_HIP, "D2H infinity flags");
        else {
            b->inf_listed = true;
            for (size_t i = 0; i < n_flags && b->inf_listed; i++)
                if (flags[i]) {
                    if (b->inf_idx.size() == czk_bases::INF_LIST_MAX) {
                        b->inf_listed = false;
                        b->inf_idx.clear();
                    } else b->inf_idx.push_back((uint32_t)i);
                }
        }
    } else if (rc == CZK_OK) b->inf_listed = n == 0;
    if (rc != CZK_OK) {
        czk_bases_release(b);
        return rc;
    }
    *out = b;
    return CZK_OK;
}

extern "C" void czk_bases_release(czk_bases* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->pts) (void)hipFree(b->pts);
    if (b->inf) (void)hipFree(b->inf);
    if (b->pts_sw0) (void)hipFree(b->pts_sw0);
    for (int i = 0; i < b->n_extra.load(); i++) {
        if (b->extra[i].pts) (void)hipFree(b->extra[i].pts);
        if (b->extra[i].inf) (void)hipFree(b->extra[i].inf);
    }
    delete b;
}

extern "C" size_t czk_bases_len(const czk_bases* b) { return b ? b->n : 0; }
extern "C" int czk_bases_layout(const czk_bases* b, unsigned* c, unsigned* windows) {
    if (!b) return CZK_ERR_ARG;
    if (c) *c = b->c;
    if (windows) *windows = b->W;
    return CZK_OK;
}

extern "C" int czk_bases_layout_for(const czk_bases* b, size_t n_scalars, unsigned* c, unsigned* windows) {
    if (!b) return CZK_ERR_ARG;
    const size_t size = b->n < n_scalars ? b->n : n_scalars;
    unsigned cc = b->c;
    if (b->split) {
        cc = choose_c_split(size);
    } else if (b->per_call_width && size) {
        const unsigned k = width_class(size);
        if (k < b->c && msm_cost(b->c, size) >= 1.12 * msm_cost(k, size)) cc = k;
    }
    if (c) *c = cc;
    if (windows) *windows = num_windows(cc);
    return CZK_OK;
}
extern "C" int czk_bases_check_subgroup(czk_ctx* ctx, const czk_bases* b, size_t* out_bad) {
    if (!ctx || !b || !out_bad) return ctx ? set_err(ctx, CZK_ERR_ARG, "null check_subgroup argument") : CZK_ERR_ARG;
    if (b->checked) {   // CZK_MEM_CHECK_SUBGROUP ran at registration
        *out_bad = b->n_bad;
        return CZK_OK;
    }
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t aw = b->group == CZK_G1 ? 12 : 24;
    const u64* pts = b->te ? b->pts_sw0 : b->pts;   // window 0 = the registered points
    u64* tmp = nullptr;
    if (b->unsat && !b->te && b->n) {   // the table is in the unsaturated residue system: check a converted copy
        CZK_HIP(ctx, hipMalloc(&tmp, b->n * aw * 8));
        hipError_t e = hipMemcpyAsync(tmp, b->pts, b->n * aw * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) {
            launch_convert_from_u(ctx->stream, tmp, b->n * (aw / 6));
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return set_err(ctx, CZK_ERR_HIP, std::string("subgroup check copy: ") + hipGetErrorString(e));
        }
        pts = tmp;
    }
    int rc = b->group == CZK_G1 ? subgroup_check_impl<Fq>(ctx, pts, b->inf, b->n, out_bad) : subgroup_check_impl<Fq2>(ctx, pts, b->inf, b->n, out_bad);
    if (tmp) (void)hipFree(tmp);
    return rc;
}
extern "C" int czk_bases_arith(const czk_bases* b) { return !b ? -1 : b->te ? 2 : b->unsat ? 1 : 0; }
extern "C" int czk_bases_prepare(czk_ctx* ctx, const czk_bases* b, size_t n_scalars) {
    if (!ctx || !b) return ctx ? set_err(ctx, CZK_ERR_ARG, "null bases") : CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t size = b->n < n_scalars ? b->n : n_scalars;
    TableView tv;
    const_cast<czk_bases*>(b)->nomem_class.store(0, std::memory_order_release);   // an explicit request retries a set that did not fit earlier
    return b->group == CZK_G1 ? pick_tables<Fq>(ctx, b, size, &tv, true) : pick_tables<Fq2>(ctx, b, size, &tv, true);
}

static int msm_common(czk_ctx* ctx, const czk_bases* bases, const uint64_t* scalars, size_t n_scalars, size_t lanes, int scalar_form, int mem,
                      uint64_t* out_jac, bool blocking) {
    if (!ctx || !bases || !out_jac) return ctx ? set_err(ctx, CZK_ERR_ARG, "null msm argument") : CZK_ERR_ARG;
    if (n_scalars && !scalars) return set_err(ctx, CZK_ERR_ARG, "null scalars");
    if (scalar_form != CZK_SCALAR_CANONICAL && scalar_form != CZK_SCALAR_MONTGOMERY) return set_err(ctx, CZK_ERR_ARG, "bad scalar_form");
    if (!lanes) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const bool stable = (mem & CZK_MEM_STABLE) != 0, same = (mem & CZK_MEM_SAME_SCALARS) != 0;
    mem &= ~(CZK_MEM_STABLE | CZK_MEM_SAME_SCALARS);
    if (!valid_mem(mem) || (stable && (blocking || mem != CZK_MEM_DEVICE)) || (same && !stable))
        return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE (CZK_MEM_STABLE: czk_msm_async with device scalars only; CZK_MEM_SAME_SCALARS: with CZK_MEM_STABLE only)");
    const u64* sdev = scalars;
    DeviceBuf tmp;
    if (mem == CZK_MEM_HOST && n_scalars) {
        CZK_TRY(stage_take(ctx, lanes * n_scalars * 32, &tmp));   // pooled: hipMalloc / hipFree per call cost more than the copy
        hipError_t e = hipMemcpyAsync(tmp.p, scalars, lanes * n_scalars * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            stage_give(ctx, tmp);
            return set_err(ctx, CZK_ERR_HIP, "H2D scalars");
        }
        sdev = (const u64*)tmp.p;
    }
    // host scalars: always blocking (the digits have been extracted from the staging buffer by the time this returns)
    int rc = msm_device(ctx, bases, sdev, n_scalars, lanes, scalar_form, out_jac, blocking || tmp.p != nullptr, stable && !blocking, same);
    if (tmp.p) stage_give(ctx, tmp);
    return rc;
}
extern "C" int czk_msm(czk_ctx* ctx, const czk_bases* bases, const uint64_t* scalars, size_t n_scalars, size_t lanes, int scalar_form, int mem,
                       uint64_t* out_jac) {
    return msm_common(ctx, bases, scalars, n_scalars, lanes, scalar_form, mem, out_jac, true);
}
extern "C" int czk_msm_async(czk_ctx* ctx, const czk_bases* bases, const uint64_t* scalars, size_t n_scalars, size_t lanes, int scalar_form,
                             int mem, uint64_t* out_jac) {
    return msm_common(ctx, bases, scalars, n_scalars, lanes, scalar_form, mem, out_jac, false);
}
extern "C" int czk_ctx_mark(czk_ctx* ctx, uint64_t* out_mark) {
    if (!ctx || !out_mark) return ctx ? set_err(ctx, CZK_ERR_ARG, "null czk_ctx_mark argument") : CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    return ctx_mark(ctx, out_mark);
}
extern "C" int czk_ctx_wait_mark(czk_ctx* ctx, uint64_t mark) {
    if (!ctx) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    return ctx_wait_mark(ctx, mark);
}

static int msm_oneshot(czk_ctx* ctx, int group, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, size_t lanes,
                       int scalar_form, uint64_t* out_jac) {
    czk_bases* b = nullptr;
    // used once: no window tables (building them costs ~20 point doublings per point and window -- more than the MSM itself)
    // ... and CZK_MEM_ANY_POINTS: this is the reference's own signature (VariableBaseMSM::multi_scalar_mul, complete on every curve point), so the
    // one-shot entry points make no subgroup assumption; registered keys (czk_bases_register) choose it themselves
    CZK_TRY(czk_bases_register(ctx, group, bases_xy, inf, n, CZK_MEM_HOST | CZK_MEM_NO_TABLES | CZK_MEM_ANY_POINTS, &b));
    int rc = czk_msm(ctx, b, scalars, n, lanes, scalar_form, CZK_MEM_HOST, out_jac);
    czk_bases_release(b);
    return rc;
}
extern "C" int czk_msm_g1(czk_ctx* ctx, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, size_t lanes,
                          int scalar_form, uint64_t* out_jac) {
    if (!ctx) return CZK_ERR_ARG;
    return msm_oneshot(ctx, CZK_G1, bases_xy, inf, scalars, n, lanes, scalar_form, out_jac);
}
extern "C" int czk_msm_g2(czk_ctx* ctx, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, size_t lanes,
                          int scalar_form, uint64_t* out_jac) {
    if (!ctx) return CZK_ERR_ARG;
    return msm_oneshot(ctx, CZK_G2, bases_xy, inf, scalars, n, lanes, scalar_form, out_jac);
}

extern "C" int czk_fixed_base_points(czk_ctx* ctx, int group, const uint64_t* k, size_t n, uint64_t* out, int mem) {
    if (!ctx || (n && (!k || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null fixed_base argument") : CZK_ERR_ARG;
    if (group != CZK_G1 && group != CZK_G2) return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t aw = group == CZK_G1 ? 12 : 24;
    if (mem == CZK_MEM_DEVICE) return fixed_base_points_device(ctx, group, k, n, out);
    void *kd = nullptr, *od = nullptr;
    CZK_HIP(ctx, hipMalloc(&kd, n * 32));
    CZK_HIP(ctx, hipMalloc(&od, n * aw * 8));
    CZK_HIP(ctx, hipMemcpyAsync(kd, k, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int rc = fixed_base_points_device(ctx, group, (const u64*)kd, n, (u64*)od);
    if (rc == CZK_OK) {
        CZK_HIP(ctx, hipMemcpyAsync(out, od, n * aw * 8, hipMemcpyDeviceToHost, ctx->stream));
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    (void)hipFree(kd);
    (void)hipFree(od);
    return rc;
}

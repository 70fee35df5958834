// ntt_pass.hip -- the NTT passes for domains of 2^11 and more: radix-8 / radix-4 butterfly groups held in registers,
// unsaturated Fr arithmetic (fru.h), two LDS exchanges per 7-stage pass instead of seven.
//
// Same transform, same values as ntt.hip's first-generation kernels (which stay for the small domains) and as the
// reference's io_helper / oi_helper + derange (algebra/poly/src/domain/radix2/fft.rs:140-260): decimation in frequency over
// the stage bits from high to low, natural order in and out.
//   * a pass owns stage bits [s_lo, s_lo + K), K in {5, 6, 7}; its 2^K x 16 tile is processed in steps of 3 or 2 stages
//     (K = 7: 3 + 2 + 2, K = 6: 3 + 3, K = 5: 3 + 2).  In a step every thread holds 8 elements in registers (one radix-8 group
//     or two radix-4 groups) and runs the step's butterflies there; the first step reads HBM directly, the last writes HBM
//     directly, steps in between exchange through LDS (9 limb planes of 2048 words: conflict-free 4-byte accesses).
//   * butterflies are lazy: lo + hi is nine adds, lo - hi adds a redundant multiple of r (fru.h); the multiply is the
//     unsaturated 153-mad Montgomery product against twiddles stored as w 2^261 mod r, so the data never changes its residue
//     system; limbs are re-normalised only where the static bound analysis below requires it (6 times per 12 butterflies).
//   * value bounds (units of r) through a pass: inputs <= 1.003 (canonical, or one multiply output after the coset
//     pre-scale); a step of m stages multiplies the bound by 2^m: 8.03 -> 32.1 -> 128.4 < 256 = what fru_canon accepts.
//     Subtractions use K r with K the next power of two above the subtrahend's bound, so a multiplier operand never exceeds
//     64.2 + 128 = 192.2 < 220 (fru_mul's limit for an output below 2 r).
//   * every value that leaves a pass is reduced to [0, r) (fru_canon): outputs are bit-identical to the reference's.  (With
//     NTT2_LAZY_SCRATCH the scratch lanes between passes hold lazy 9-limb elements < 2 r instead; such a pass starts from
//     the bound 2 and uses the next larger constants: 16 -> 64 -> 256, multiplier operand <= 128 + 256 = 384 < 440, output
//     still below 2 r because twiddles are canonical.  Measured slower, off by default: see ntt_pass.h.)
#include "ntt_pass.h"

namespace czk {

__device__ __forceinline__ FrU tab_load(const u32* tab, size_t idx) {
    FrU r;
    const u32* p = tab + 9 * idx;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}

// DIF butterfly: (a, b) <- (a + b, (a - b + K r) w)
template <int K, int U>
__device__ __forceinline__ void bfly(FrU& a, FrU& b, const FrU& w) {
    FrU d = fru_sub<K, U>(a, b);
    a = fru_add(a, b);
    b = fru_mul(d, w);
}

// Three DIF stages on x[0..8) (x[k]: k = the three stage bits, most significant = first stage).  Inputs normalised, value
// < KB r.  tw[0..4): first stage, pair (k, k + 4); tw[4..6): second stage, pairs (k, k + 2), k in {0, 1} (both halves);
// tw[6]: third stage.  Outputs normalised: x[0] < 8 KB r, the others smaller.
template <int KB>
__device__ __forceinline__ void radix8(FrU* x, const FrU* tw) {
#pragma unroll
    for (int k = 0; k < 4; k++) bfly<KB, 1>(x[k], x[k + 4], tw[k]);                 // sums: limbs < 2^30, < 2 KB r; products: multiply outputs
    bfly<2 * KB, 2>(x[0], x[2], tw[4]);                                              // operands with limbs < 2^30
    bfly<2 * KB, 2>(x[1], x[3], tw[5]);
    bfly<2, 1>(x[4], x[6], tw[4]);                                                   // operands are multiply outputs (< 2 r, limbs < 2^29)
    bfly<2, 1>(x[5], x[7], tw[5]);
    x[0] = fru_normalize(x[0]);                                                      // limbs < 2^31 after two lazy sums
    x[1] = fru_normalize(x[1]);
    bfly<4 * KB, 1>(x[0], x[1], tw[6]);
    bfly<2, 1>(x[2], x[3], tw[6]);
    bfly<4, 2>(x[4], x[5], tw[6]);                                                   // sums of two multiply outputs: limbs < 2^30, < 4 r
    bfly<2, 1>(x[6], x[7], tw[6]);
    x[0] = fru_normalize(x[0]);
    x[2] = fru_normalize(x[2]);
    x[4] = fru_normalize(x[4]);
    x[6] = fru_normalize(x[6]);
}
// Two DIF stages on x[0..4).  tw[0..2): first stage, pair (k, k + 2); tw[2]: second stage.
template <int KB>
__device__ __forceinline__ void radix4(FrU* x, const FrU* tw) {
    bfly<KB, 1>(x[0], x[2], tw[0]);
    bfly<KB, 1>(x[1], x[3], tw[1]);
    bfly<2 * KB, 2>(x[0], x[1], tw[2]);
    bfly<2, 1>(x[2], x[3], tw[2]);
    x[0] = fru_normalize(x[0]);
    x[2] = fru_normalize(x[2]);
}

// The LAST step of a transform (stage bits 1, 0 or 2, 1, 0): the twiddle of stage 0 is w^0 = 1 and so is every stage's entry
// j = 0, i.e. 3 of the 4 (7 of the 12) multiplies of a group would multiply by one.  They are left out: the difference stays a
// lazy value (no Montgomery product to bring it below 2 r), so ALL outputs are re-normalised instead of half of them, and the
// bounds become (units of r, KB = bound of the inputs, real inputs <= KB / 2 + 0.1):
//   radix-4: d02 < 1.5 KB;  x0 < 2 KB;  x1 = s02 - x1 + 2 KB r < 3 KB + 0.2;  x2, x3 < 1.5 KB + 2      -> max 192.2 at KB = 64, 384.2 at
//   KB = 128 (inputs < 2 r: scratch formats 1 and 2);  radix-8: x1 = x0 - x1 + 4 KB r < 6 KB + 0.1 = 96.1 at KB = 16, 192.1 at KB = 32.
//   Limits: 440 for fru_mul against a canonical factor (the post-scale constants are), 2^261 = 438 r for the limb form and for
//   fru_canon's quotient estimate.  Limbs stay below 3.5 * 2^30 before the normalisation.
// tw1 = the stage-1 twiddle of index 1 (the primitive fourth root); tw2[1..4) = stage-2 twiddles of index 1..3.
template <int KB>
__device__ __forceinline__ void radix4_last(FrU* x, const FrU& tw1) {
    const FrU d02 = fru_sub<KB, 1>(x[0], x[2]);
    const FrU s02 = fru_add(x[0], x[2]);
    bfly<KB, 1>(x[1], x[3], tw1);
    x[0] = fru_normalize(fru_add(s02, x[1]));
    x[1] = fru_normalize(fru_sub<2 * KB, 2>(s02, x[1]));
    x[2] = fru_normalize(fru_add(d02, x[3]));
    x[3] = fru_normalize(fru_sub<2, 1>(d02, x[3]));
}
template <int KB>
__device__ __forceinline__ void radix8_last(FrU* x, const FrU* tw2, const FrU& tw1) {
    const FrU d04 = fru_sub<KB, 1>(x[0], x[4]);
    x[0] = fru_add(x[0], x[4]);
#pragma unroll
    for (int k = 1; k < 4; k++) bfly<KB, 1>(x[k], x[k + 4], tw2[k]);
    // stage 1: pairs (0, 2) and (4, 6) have the unit twiddle
    const FrU d = fru_sub<2 * KB, 2>(x[0], x[2]);
    x[0] = fru_add(x[0], x[2]);
    x[2] = d;
    bfly<2 * KB, 2>(x[1], x[3], tw1);
    const FrU e = fru_sub<2, 1>(d04, x[6]);
    x[4] = fru_add(d04, x[6]);
    x[6] = e;
    bfly<2, 1>(x[5], x[7], tw1);
    // stage 0: unit twiddles only
    x[0] = fru_normalize(x[0]);
    x[1] = fru_normalize(x[1]);
    const FrU s01 = fru_add(x[0], x[1]);
    x[1] = fru_normalize(fru_sub<4 * KB, 1>(x[0], x[1]));
    x[0] = fru_normalize(s01);
    const FrU s23 = fru_add(x[2], x[3]);
    x[3] = fru_normalize(fru_sub<2, 1>(x[2], x[3]));
    x[2] = fru_normalize(s23);
    const FrU s45 = fru_add(x[4], x[5]);
    x[5] = fru_normalize(fru_sub<4, 2>(x[4], x[5]));
    x[4] = fru_normalize(s45);
    const FrU s67 = fru_add(x[6], x[7]);
    x[7] = fru_normalize(fru_sub<2, 1>(x[6], x[7]));
    x[6] = fru_normalize(s67);
}

constexpr unsigned NTT2_LOGT = 4, NTT2_T = 16;

// LDS: 9 limb planes of NE words
template <int NE>
__device__ __forceinline__ FrU lds_get9(const u32* s, unsigned e) {
    FrU r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = s[i * NE + e];
    return r;
}
template <int NE>
__device__ __forceinline__ void lds_put9(u32* s, unsigned e, const FrU& v) {
#pragma unroll
    for (int i = 0; i < 9; i++) s[i * NE + e] = v.l[i];
}

template <bool LAZY_IN>
__device__ __forceinline__ FrU load_input(const Pass2Args& a, const u64* in, size_t gi) {
    if constexpr (LAZY_IN && NTT2_LAZY_SCRATCH) return tab_load((const u32*)in, gi);   // 36-byte scratch lanes: 9 limbs per element
    // (scratch format 2: the same 8 x u32 re-slicing as a canonical element, the value is merely < 2 r instead of < r)
    if (a.first && gi >= a.in_len) {
        FrU z;
#pragma unroll
        for (int i = 0; i < 9; i++) z.l[i] = 0;
        return z;
    }
    FrU v = fru_unpack(fp_load<FrParams>(in + 4 * gi));
    if (a.prescale) v = fru_mul(v, tab_load(a.prescale, gi));   // distribute_powers(g) on the un-resized input (domain/mod.rs:93-106, 139-142)
    return v;
}
__device__ __forceinline__ void store_output(const Pass2Args& a, u64* out, size_t k, const FrU& v, bool last) {
    if (!last && NTT2_SCRATCH_2R && !NTT2_LAZY_SCRATCH) {   // scratch format 2: < 2 r, packed; no conditional subtraction
        fp_store<FrParams>(out + 4 * k, fru_pack(fru_reduce_2r(v)));
        return;
    }
    if (!last && NTT2_LAZY_SCRATCH) {   // scratch lanes, lazy format
        const FrU t = fru_reduce_2r(v);
        u32* p = (u32*)out + 9 * k;
#pragma unroll
        for (int i = 0; i < 9; i++) p[i] = t.l[i];
        return;
    }
    Fr s;
    if (last && a.post_mode == 1) s = fru_canon_mulout(fru_mul(v, a.postconst));               // * size_inv (fft.rs:26-29)
    else if (last && a.post_mode == 2) s = fru_canon_mulout(fru_mul(v, tab_load(a.posttab, k)));   // * size_inv g^-i (domain/mod.rs:99-106)
    else s = fru_canon(v);
    fp_store<FrParams>(out + 4 * k, s);
}

// ------------------------------------------------------------------------------------------------------------
// strided pass: stage bits [s_lo, s_lo + K), s_lo >= 4.  Tile: rows r (K bits) x 16 consecutive columns t.
// ------------------------------------------------------------------------------------------------------------
// one step over the row bits [B0, B0 + RB) of the tile
template <int K, int B0, int RB, int KB, bool FROM_GLOBAL, bool TO_GLOBAL, bool LAZY_IN>
__device__ __forceinline__ void strided_step(const Pass2Args& a, u32* smem, const u64* in, u64* out, size_t base, unsigned low0) {
    constexpr int NE = (1 << K) * NTT2_T, R = 1 << RB, GROUPS = 8 / R, NTHR = NE / 8;
    const unsigned u = threadIdx.x;
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
        const unsigned gid = u + g * NTHR;
        const unsigned t = gid & (NTT2_T - 1), ro = gid >> NTT2_LOGT;
        const unsigned r_below = ro & ((1u << B0) - 1u), r_above = ro >> B0;
        const unsigned r0 = (r_above << (B0 + RB)) | r_below;
        FrU x[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned r = r0 | ((unsigned)k << B0);
            if constexpr (FROM_GLOBAL) x[k] = load_input<LAZY_IN>(a, in, base | ((size_t)r << a.s_lo) | t);
            else x[k] = lds_get9<NE>(smem, r * NTT2_T + t);
        }
        // twiddles: stage bit q of the tile = stage s_lo + q of the transform; index j = element index mod 2^s of the
        // butterfly's low element = (rlow << s_lo) | low0 | t with rlow = r mod 2^q
        const size_t col = (size_t)low0 | t;
        if constexpr (RB == 3) {
            FrU tw[7];
            {
                const unsigned q = B0 + 2, s = a.s_lo + q;
                const u32* tws = a.tw + 9 * (((size_t)1 << s) - 1);
#pragma unroll
                for (int k = 0; k < 4; k++) tw[k] = tab_load(tws, (((size_t)(r_below | ((unsigned)k << B0))) << a.s_lo) | col);
            }
            {
                const unsigned q = B0 + 1, s = a.s_lo + q;
                const u32* tws = a.tw + 9 * (((size_t)1 << s) - 1);
#pragma unroll
                for (int k = 0; k < 2; k++) tw[4 + k] = tab_load(tws, (((size_t)(r_below | ((unsigned)k << B0))) << a.s_lo) | col);
            }
            {
                const unsigned s = a.s_lo + B0;
                tw[6] = tab_load(a.tw + 9 * (((size_t)1 << s) - 1), ((size_t)r_below << a.s_lo) | col);
            }
            radix8<KB>(x, tw);
        } else {
            FrU tw[3];
            {
                const unsigned q = B0 + 1, s = a.s_lo + q;
                const u32* tws = a.tw + 9 * (((size_t)1 << s) - 1);
#pragma unroll
                for (int k = 0; k < 2; k++) tw[k] = tab_load(tws, (((size_t)(r_below | ((unsigned)k << B0))) << a.s_lo) | col);
            }
            {
                const unsigned s = a.s_lo + B0;
                tw[2] = tab_load(a.tw + 9 * (((size_t)1 << s) - 1), ((size_t)r_below << a.s_lo) | col);
            }
            radix4<KB>(x, tw);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned r = r0 | ((unsigned)k << B0);
            if constexpr (TO_GLOBAL) store_output(a, out, base | ((size_t)r << a.s_lo) | t, x[k], false);
            else lds_put9<NE>(smem, r * NTT2_T + t, x[k]);
        }
    }
}

template <int K, bool LAZY_IN>
__global__ __launch_bounds__((1 << K) * 2) void k_ntt2_strided(Pass2Args a) {
    extern __shared__ __attribute__((aligned(16))) u32 smem2[];
    const unsigned lb_bits = a.s_lo - NTT2_LOGT;
    const unsigned B = blockIdx.x;
    const unsigned Lb = B & ((1u << lb_bits) - 1u);
    const size_t H = B >> lb_bits;
    const size_t base = (H << (a.s_lo + K)) | ((size_t)Lb << NTT2_LOGT);
    const unsigned low0 = Lb << NTT2_LOGT;
    constexpr int W = LAZY_IN ? 2 : 1;   // inputs < 2 r instead of < 1.003 r: constants one power of two up
    const u64* in = (const u64*)((const char*)a.in + (a.first ? 32 * a.in_lane_stride : (LAZY_IN ? NTT2_SCRATCH_ELEM_BYTES : 32) * a.lane_stride) * blockIdx.y);
    u64* out = (u64*)((char*)a.out + NTT2_SCRATCH_ELEM_BYTES * a.lane_stride * blockIdx.y);
    if constexpr (K == 7) {
        strided_step<7, 4, 3, 2 * W, true, false, LAZY_IN>(a, smem2, in, out, base, low0);
        __syncthreads();
        strided_step<7, 2, 2, 16 * W, false, false, LAZY_IN>(a, smem2, in, out, base, low0);
        __syncthreads();
        strided_step<7, 0, 2, 64 * W, false, true, LAZY_IN>(a, smem2, in, out, base, low0);
    } else if constexpr (K == 6) {
        strided_step<6, 3, 3, 2 * W, true, false, LAZY_IN>(a, smem2, in, out, base, low0);
        __syncthreads();
        strided_step<6, 0, 3, 16 * W, false, true, LAZY_IN>(a, smem2, in, out, base, low0);
    } else {
        strided_step<5, 2, 3, 2 * W, true, false, LAZY_IN>(a, smem2, in, out, base, low0);
        __syncthreads();
        strided_step<5, 0, 2, 16 * W, false, true, LAZY_IN>(a, smem2, in, out, base, low0);
    }
}

// ------------------------------------------------------------------------------------------------------------
// last pass: stage bits [0, C); a block owns 16 contiguous chunks of 2^C whose bit-reversed chunk ids are consecutive, so
// the transposed store is the bit-reversal permutation (derange, fft.rs:253-260) and still writes 512-byte runs.
// LDS: chunk t at row stride RS = 2^C + 1 words per limb plane.
// ------------------------------------------------------------------------------------------------------------
template <int C, int B0, int RB, int KB, bool FROM_GLOBAL, bool TO_GLOBAL>
__device__ __forceinline__ void final_step(const Pass2Args& a, u32* smem, const u64* in, u64* out, unsigned blk) {
    constexpr int LEN = 1 << C, RS = LEN + 1, NEL = RS * NTT2_T, R = 1 << RB, GROUPS = 8 / R, NTHR = LEN * NTT2_T / 8;
    constexpr int OB = C - RB;   // bits of l outside the step
    const unsigned hb = a.n - C;
    const unsigned u = threadIdx.x;
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
        const unsigned gid = u + g * NTHR;
        unsigned t, lo;
        if constexpr (TO_GLOBAL) {   // t fastest: the transposed store writes 16 consecutive outputs per 16 threads
            t = gid & (NTT2_T - 1);
            lo = gid >> NTT2_LOGT;
        } else {                      // l fastest: contiguous loads, conflict-free LDS rows
            lo = gid & ((1u << OB) - 1u);
            t = gid >> OB;
        }
        const unsigned l_below = lo & ((1u << B0) - 1u), l_above = lo >> B0;
        const unsigned l0 = (l_above << (B0 + RB)) | l_below;
        const size_t chunk = (size_t)blk * NTT2_T + t;
        const size_t h = hb ? (size_t)(__brev((unsigned)chunk) >> (32 - hb)) : 0;
        FrU x[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned l = l0 | ((unsigned)k << B0);
            if constexpr (FROM_GLOBAL) x[k] = load_input<NTT2_SCRATCH_2R>(a, in, (h << C) | l);
            else x[k] = lds_get9<NEL>(smem, t * RS + l);
        }
        if constexpr (B0 == 0) {   // the transform's last stages: unit twiddles are not multiplied (radix*_last)
            const u32* tw_s1 = a.tw + 9 * 1;             // stage 1: entries 1, 2 of the compacted table = w_4^0, w_4^1
            if constexpr (RB == 3) {
                FrU tw2[4];
#pragma unroll
                for (int k = 1; k < 4; k++) tw2[k] = tab_load(a.tw + 9 * 3, k);
                radix8_last<KB>(x, tw2, tab_load(tw_s1, 1));
            } else {
                radix4_last<KB>(x, tab_load(tw_s1, 1));
            }
        } else if constexpr (RB == 3) {
            FrU tw[7];
#pragma unroll
            for (int k = 0; k < 4; k++) tw[k] = tab_load(a.tw + 9 * (((size_t)1 << (B0 + 2)) - 1), l_below | ((unsigned)k << B0));
#pragma unroll
            for (int k = 0; k < 2; k++) tw[4 + k] = tab_load(a.tw + 9 * (((size_t)1 << (B0 + 1)) - 1), l_below | ((unsigned)k << B0));
            tw[6] = tab_load(a.tw + 9 * (((size_t)1 << B0) - 1), l_below);
            radix8<KB>(x, tw);
        } else {
            FrU tw[3];
#pragma unroll
            for (int k = 0; k < 2; k++) tw[k] = tab_load(a.tw + 9 * (((size_t)1 << (B0 + 1)) - 1), l_below | ((unsigned)k << B0));
            tw[2] = tab_load(a.tw + 9 * (((size_t)1 << B0) - 1), l_below);
            radix4<KB>(x, tw);
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned l = l0 | ((unsigned)k << B0);
            if constexpr (TO_GLOBAL) {
                const size_t ko = ((size_t)(__brev(l) >> (32 - C)) << hb) | chunk;
                store_output(a, out, ko, x[k], true);
            } else {
                lds_put9<NEL>(smem, t * RS + l, x[k]);
            }
        }
    }
}

template <int C>
__global__ __launch_bounds__((1 << C) * 2) void k_ntt2_final(Pass2Args a) {   // input: scratch lanes
    extern __shared__ __attribute__((aligned(16))) u32 smem2[];
    constexpr int W = NTT2_SCRATCH_2R ? 2 : 1;   // inputs < 2 r
    const u64* in = (const u64*)((const char*)a.in + NTT2_SCRATCH_ELEM_BYTES * a.lane_stride * blockIdx.y);
    u64* out = a.out + 4 * a.lane_stride * blockIdx.y;
    if constexpr (C == 7) {
        final_step<7, 4, 3, 2 * W, true, false>(a, smem2, in, out, blockIdx.x);
        __syncthreads();
        final_step<7, 2, 2, 16 * W, false, false>(a, smem2, in, out, blockIdx.x);
        __syncthreads();
        final_step<7, 0, 2, 64 * W, false, true>(a, smem2, in, out, blockIdx.x);
    } else if constexpr (C == 6) {
        final_step<6, 3, 3, 2 * W, true, false>(a, smem2, in, out, blockIdx.x);
        __syncthreads();
        final_step<6, 0, 3, 16 * W, false, true>(a, smem2, in, out, blockIdx.x);
    } else {
        final_step<5, 2, 3, 2 * W, true, false>(a, smem2, in, out, blockIdx.x);
        __syncthreads();
        final_step<5, 0, 2, 16 * W, false, true>(a, smem2, in, out, blockIdx.x);
    }
}

// ------------------------------------------------------------------------------------------------------------
// fused pass of an `ifft -> coset_fft` pair (R1CStoQAP::witness_map, mpc-snarks/src/groth/r1cs_to_qap.rs:85-89, 102-103): the LAST pass of the inverse
// transform (stage bits [0, C), bit-reversing store) and the FIRST pass of the forward transform (stage bits [n - C, n)) own the same tile -- block
// `blk` of the last pass writes natural indices (brev_C(l) << (n - C)) | (16 blk + t), which is tile Lb = blk of the strided pass with row r = brev_C(l)
// -- so the tile makes one trip through HBM instead of two.  Every value takes the steps it takes in the two kernels (canonical after the inverse
// transform's 1 / D, times g^i, then the forward butterflies), so the results are the unfused ones bit for bit.
// ------------------------------------------------------------------------------------------------------------
// the last step of the inverse transform with its outputs kept in registers: y[g * R + k] = the forward transform's input element
template <int C, int RB, int KB>
__device__ __forceinline__ void final_step_keep(const Pass2Args& ai, const Pass2Args& af, const u32* smem, unsigned blk, FrU* y) {
    constexpr int LEN = 1 << C, RS = LEN + 1, NEL = RS * NTT2_T, R = 1 << RB, GROUPS = 8 / R, NTHR = LEN * NTT2_T / 8;
    const unsigned hb = ai.n - C;
    const unsigned u = threadIdx.x;
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
        const unsigned gid = u + g * NTHR;
        const unsigned t = gid & (NTT2_T - 1), lo = gid >> NTT2_LOGT;
        const unsigned l0 = lo << RB;   // B0 == 0: the step's bits are the lowest
        const size_t chunk = (size_t)blk * NTT2_T + t;
        FrU x[R];
#pragma unroll
        for (int k = 0; k < R; k++) x[k] = lds_get9<NEL>(smem, t * RS + (l0 | (unsigned)k));
        const u32* tw_s1 = ai.tw + 9 * 1;
        if constexpr (RB == 3) {
            FrU tw2[4];
#pragma unroll
            for (int k = 1; k < 4; k++) tw2[k] = tab_load(ai.tw + 9 * 3, k);
            radix8_last<KB>(x, tw2, tab_load(tw_s1, 1));
        } else {
            radix4_last<KB>(x, tab_load(tw_s1, 1));
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned l = l0 | (unsigned)k;
            const size_t ko = ((size_t)(__brev(l) >> (32 - C)) << hb) | chunk;
            const Fr s = fru_canon_mulout(fru_mul(x[k], ai.postconst));          // what the inverse transform stores: x / D, canonical (fft.rs:26-29)
            y[g * R + k] = fru_mul(fru_unpack(s), tab_load(af.prescale, ko));    // what the forward transform loads: times g^i (domain/mod.rs:93-106)
        }
    }
}
// ... and their hand-over into the strided pass's LDS layout (row r = brev_C(l), column t)
template <int C, int RB>
__device__ __forceinline__ void final_put_strided(u32* smem, const FrU* y) {
    constexpr int LEN = 1 << C, NE = LEN * NTT2_T, R = 1 << RB, GROUPS = 8 / R, NTHR = LEN * NTT2_T / 8;
    const unsigned u = threadIdx.x;
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
        const unsigned gid = u + g * NTHR;
        const unsigned t = gid & (NTT2_T - 1), l0 = (gid >> NTT2_LOGT) << RB;
#pragma unroll
        for (int k = 0; k < R; k++) lds_put9<NE>(smem, (__brev(l0 | (unsigned)k) >> (32 - C)) * NTT2_T + t, y[g * R + k]);
    }
}
template <int C>
__global__ __launch_bounds__((1 << C) * 2) void k_ntt2_final_first(Pass2Args ai, Pass2Args af) {
    extern __shared__ __attribute__((aligned(16))) u32 smem2[];
    constexpr int W = NTT2_SCRATCH_2R ? 2 : 1;   // the inverse transform's scratch lanes: inputs < 2 r
    const u64* in = (const u64*)((const char*)ai.in + NTT2_SCRATCH_ELEM_BYTES * ai.lane_stride * blockIdx.y);
    u64* out = (u64*)((char*)af.out + NTT2_SCRATCH_ELEM_BYTES * af.lane_stride * blockIdx.y);
    const unsigned blk = blockIdx.x;
    const size_t base = (size_t)blk << NTT2_LOGT;   // s_lo + C == n: one tile row of blocks
    const unsigned low0 = blk << NTT2_LOGT;
    FrU y[8];
    if constexpr (C == 7) {
        final_step<7, 4, 3, 2 * W, true, false>(ai, smem2, in, nullptr, blk);
        __syncthreads();
        final_step<7, 2, 2, 16 * W, false, false>(ai, smem2, in, nullptr, blk);
        __syncthreads();
        final_step_keep<7, 2, 64 * W>(ai, af, smem2, blk, y);
        __syncthreads();
        final_put_strided<7, 2>(smem2, y);
        __syncthreads();
        strided_step<7, 4, 3, 2, false, false, false>(af, smem2, nullptr, out, base, low0);
        __syncthreads();
        strided_step<7, 2, 2, 16, false, false, false>(af, smem2, nullptr, out, base, low0);
        __syncthreads();
        strided_step<7, 0, 2, 64, false, true, false>(af, smem2, nullptr, out, base, low0);
    } else if constexpr (C == 6) {
        final_step<6, 3, 3, 2 * W, true, false>(ai, smem2, in, nullptr, blk);
        __syncthreads();
        final_step_keep<6, 3, 16 * W>(ai, af, smem2, blk, y);
        __syncthreads();
        final_put_strided<6, 3>(smem2, y);
        __syncthreads();
        strided_step<6, 3, 3, 2, false, false, false>(af, smem2, nullptr, out, base, low0);
        __syncthreads();
        strided_step<6, 0, 3, 16, false, true, false>(af, smem2, nullptr, out, base, low0);
    } else {
        final_step<5, 2, 3, 2 * W, true, false>(ai, smem2, in, nullptr, blk);
        __syncthreads();
        final_step_keep<5, 2, 16 * W>(ai, af, smem2, blk, y);
        __syncthreads();
        final_put_strided<5, 2>(smem2, y);
        __syncthreads();
        strided_step<5, 2, 3, 2, false, false, false>(af, smem2, nullptr, out, base, low0);
        __syncthreads();
        strided_step<5, 0, 2, 16, false, true, false>(af, smem2, nullptr, out, base, low0);
    }
}

// saturated table entry w R (8 x u32, canonical) -> unsaturated w 2^261 mod r (9 x 29-bit limbs)
__global__ void k_table_to_u(const u64* sat, size_t count, u32* dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr k32;
#pragma unroll
    for (int j = 0; j < 8; j++) k32.l[j] = fru_k32(j);
    FrU v = fru_unpack(fp_mul(fp_load<FrParams>(sat + 4 * i), k32));   // 32 w R = w 2^261 (mod r)
#pragma unroll
    for (int j = 0; j < 9; j++) dst[9 * i + j] = v.l[j];
}

void launch_table_to_u(hipStream_t st, const u64* sat, size_t count, u32* dst) {
    if (!count) return;
    hipLaunchKernelGGL(k_table_to_u, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, sat, count, dst);
}

FrU host_fr_to_u(const Fr& sat) {   // host-side conversion of one constant (size_inv)
    Fr k32;
    for (int j = 0; j < 8; j++) k32.l[j] = fru_k32(j);
    Fr v = fp_mul(sat, k32);
    FrU r;
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, off = bit & 31;
        u64 two = (u64)v.l[w] | ((w + 1) < 8 ? ((u64)v.l[w + 1] << 32) : 0);
        u32 x = (u32)(two >> off);
        r.l[i] = i < 8 ? (x & FRU_MASK) : x;
    }
    return r;
}

// one pass of the second-generation NTT.  K in {5, 6, 7}; `first`: reads the caller's canonical lanes (otherwise the lazy
// scratch lanes); `last`: stage bits [0, K) with the transposed store of canonical elements.
// the fused pass above: `ai` = the inverse transform's last pass (in: its scratch lanes), `af` = the forward transform's first pass (out: the OTHER scratch
// lanes -- a block reads and writes different element sets, so the pass cannot run in place); C = K = the stage bits both own
int launch_ntt2_final_first(czk_ctx* ctx, const Pass2Args& ai, const Pass2Args& af, unsigned C, size_t lanes) {
    const size_t D = (size_t)1 << ai.n;
    const dim3 grid((unsigned)(D >> (C + NTT2_LOGT)), (unsigned)lanes), block((1u << C) * 2);
    const size_t lds = (size_t)9 * (((size_t)1 << C) + 1) * NTT2_T * 4;   // the last pass's padded rows: the larger of the two layouts
    if (lds > ctx->lds_per_block) return set_err(ctx, CZK_ERR_HIP, "fused NTT pass needs " + std::to_string(lds) + " bytes of LDS per workgroup");
    if (C == 7) hipLaunchKernelGGL(k_ntt2_final_first<7>, grid, block, lds, ctx->stream, ai, af);
    else if (C == 6) hipLaunchKernelGGL(k_ntt2_final_first<6>, grid, block, lds, ctx->stream, ai, af);
    else hipLaunchKernelGGL(k_ntt2_final_first<5>, grid, block, lds, ctx->stream, ai, af);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

int launch_ntt2_pass(czk_ctx* ctx, const Pass2Args& a, unsigned K, bool last, size_t lanes) {
    const size_t D = (size_t)1 << a.n;
    const unsigned blocks = (unsigned)(D >> (K + NTT2_LOGT));
    const dim3 grid(blocks, (unsigned)lanes), block((1u << K) * 2);
    // a 7-stage tile is 9 limb planes x 2048 words = 72 KiB (73.5 KiB with the row padding of the last pass): two workgroups per
    // CU out of gfx950's 160 KiB.  Checked against the device rather than assumed.
    const size_t lds_need = last ? (size_t)9 * (((size_t)1 << K) + 1) * NTT2_T * 4 : (size_t)9 * ((size_t)1 << (K + NTT2_LOGT)) * 4;
    if (lds_need > ctx->lds_per_block)
        return set_err(ctx, CZK_ERR_HIP, "NTT pass needs " + std::to_string(lds_need) + " bytes of LDS per workgroup; this device offers " + std::to_string(ctx->lds_per_block));
    if (!last) {
        const size_t lds = (size_t)9 * ((size_t)1 << (K + NTT2_LOGT)) * 4;
        if (a.first || !NTT2_SCRATCH_2R) {
            if (K == 7) hipLaunchKernelGGL((k_ntt2_strided<7, false>), grid, block, lds, ctx->stream, a);
            else if (K == 6) hipLaunchKernelGGL((k_ntt2_strided<6, false>), grid, block, lds, ctx->stream, a);
            else hipLaunchKernelGGL((k_ntt2_strided<5, false>), grid, block, lds, ctx->stream, a);
        } else {
            if (K == 7) hipLaunchKernelGGL((k_ntt2_strided<7, true>), grid, block, lds, ctx->stream, a);
            else if (K == 6) hipLaunchKernelGGL((k_ntt2_strided<6, true>), grid, block, lds, ctx->stream, a);
            else hipLaunchKernelGGL((k_ntt2_strided<5, true>), grid, block, lds, ctx->stream, a);
        }
    } else {
        const size_t lds = (size_t)9 * (((size_t)1 << K) + 1) * NTT2_T * 4;
        if (K == 7) hipLaunchKernelGGL(k_ntt2_final<7>, grid, block, lds, ctx->stream, a);
        else if (K == 6) hipLaunchKernelGGL(k_ntt2_final<6>, grid, block, lds, ctx->stream, a);
        else hipLaunchKernelGGL(k_ntt2_final<5>, grid, block, lds, ctx->stream, a);
    }
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

}  // namespace czk

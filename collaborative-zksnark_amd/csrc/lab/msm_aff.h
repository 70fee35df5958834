// msm_aff.h -- batched-affine pre-reduction of the sorted bucket lists (G1), in front of the XYZZ bucket accumulation.
//
// A bucket's points are summed as a binary tree: round 1 adds the points pairwise, round 2 the results pairwise, ... and the
// XYZZ kernel folds what is left (after R rounds: ceil(cnt / 2^R) points per bucket).  All additions of a round are
// independent, so each thread takes AFF_KB of them, adds them in AFFINE coordinates
//     lambda = (y2 - y1) / (x2 - x1),  x3 = lambda^2 - x1 - x2,  y3 = lambda (x1 - x3) - y1        (5 M + 1 S with the trick below)
// and shares ONE field inversion between them (Montgomery's trick: running products forward, one inversion, unwinding
// backward).  The inversion is Bernstein-Yang safegcd (fq_safegcd.h, ~35 k instructions): a wave executes it in lock step
// whether one lane needs it or all 64, so it can only be amortised over additions of the same lane -- hence AFF_KB = 64 per
// thread (measured in tools/affine_bench.hip: 3406 + 34500 / K instructions per addition against 4466 for the XYZZ mixed
// addition; profiles/r02_affine_prototype.json).  Same group elements as the reference's bucket sums
// (algebra/ec/src/msm/variable_base.rs:50-64); the exceptional cases of short_weierstrass_jacobian.rs:570-597 (equal points,
// opposite points, infinity) are detected with a one-compare filter and redone by k_affine_fix with complete formulas.
//
// Memory:
//   * records rec_r[s] = (source A, source B) for every output slot s of round r: a pair to add, a single point to copy
//     (B = AFF_NONE) or nothing (A = AFF_NONE).  Round 1 sources are table codes (index | sign << 31), later rounds slots of the
//     previous level.  Bucket b's outputs start at off_r[b] = (off_{r-1}[b] + rank(b) + 1) >> 1 (rank = position of the bucket in
//     memory order), which never overlaps the next bucket and needs no scan.
//   * level arrays hold affine points in the unsaturated residue system, 128 bytes per slot: x and y as 14 limbs + 2 spare words
//     (word 15 of x = flags).  Slots are stored in groups of 64 with the eight 16-byte chunks interleaved, so that a wave touching
//     64 consecutive slots reads and writes 1 KiB runs.
//   * a wave processes AFF_KB x 64 consecutive output slots per work item; its running products live in a private 256 KiB scratch
//     region (L2 / MALL resident).
#pragma once
#include "czk_internal.h"
#include "fq_safegcd.h"
#include "fqu.h"

namespace czk {

constexpr unsigned AFF_KB = 64;                 // additions per thread and inversion
constexpr unsigned AFF_ITEM = 64 * AFF_KB;      // output slots per wave work item
constexpr u32 AFF_NONE = 0xffffffffu;
constexpr u32 AFF_F_INF = 1u;                   // flags word (x limb slot 15)
constexpr unsigned AFF_MAX_ROUNDS = 3;

// ---- level array access ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t lvl_idx(size_t slot, int c) { return (((slot >> 6) * 8 + c) << 6) | (slot & 63); }

struct AffPoint {
    FqU x, y;
    u32 flags;
};
__device__ __forceinline__ void lvl_load_coord(const uint4* lvl, size_t slot, int c0, FqU& v, u32* w15) {
    uint4 q[4];
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = lvl[lvl_idx(slot, c0 + c)];
    v.l[0] = q[0].x; v.l[1] = q[0].y; v.l[2] = q[0].z; v.l[3] = q[0].w;
    v.l[4] = q[1].x; v.l[5] = q[1].y; v.l[6] = q[1].z; v.l[7] = q[1].w;
    v.l[8] = q[2].x; v.l[9] = q[2].y; v.l[10] = q[2].z; v.l[11] = q[2].w;
    v.l[12] = q[3].x; v.l[13] = q[3].y;
    if (w15) *w15 = q[3].w;
}
__device__ __forceinline__ void lvl_store_coord(uint4* lvl, size_t slot, int c0, const FqU& v, u32 w15) {
    lvl[lvl_idx(slot, c0)] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    lvl[lvl_idx(slot, c0 + 1)] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    lvl[lvl_idx(slot, c0 + 2)] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    lvl[lvl_idx(slot, c0 + 3)] = make_uint4(v.l[12], v.l[13], 0u, w15);
}
__device__ __forceinline__ AffPoint lvl_load(const uint4* lvl, size_t slot) {
    AffPoint p;
    lvl_load_coord(lvl, slot, 0, p.x, &p.flags);
    lvl_load_coord(lvl, slot, 4, p.y, nullptr);
    return p;
}
__device__ __forceinline__ void lvl_store(uint4* lvl, size_t slot, const AffPoint& p) {
    lvl_store_coord(lvl, slot, 0, p.x, p.flags);
    lvl_store_coord(lvl, slot, 4, p.y, 0u);
}
// table entry (x R', y R' as canonical 12 x u32; msm.hip register_impl) -> unsaturated point; a negative digit adds -P
__device__ __forceinline__ void table_load_x(const u64* pts, u32 code, FqU& x) { x = fqu_unpack(fp_load<FqParams>(pts + (size_t)12 * (code & 0x7fffffffu))); }
__device__ __forceinline__ void table_load_y(const u64* pts, u32 code, FqU& y) {
    y = fqu_unpack(fp_load<FqParams>(pts + (size_t)12 * (code & 0x7fffffffu) + 6));
    if (code & 0x80000000u) {
        FqU t;
#pragma unroll
        for (int i = 0; i < 14; i++) t.l[i] = fqu_4p(i) - y.l[i];   // 4 p - y, normalised below: limbs < 2^28, value <= 4 p
        y = fqu_normalize(t);
    }
}

// a (normalised limbs, value < 64 p) -> the same residue in [0, 3 p), normalised (quotient estimate from the top limb)
__device__ __forceinline__ FqU fqu_reduce_small(const FqU& a) {
    const u32 q = a.l[13] / 6884u;                       // p >> 364 = 6883.6: q in {floor(a / p) - 1, floor(a / p)}
    FqU r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        acc += (int64_t)a.l[i] - (int64_t)((u64)q * fqu_p(i));
        r.l[i] = (u32)acc & FQU_MASK;
        acc >>= 28;
    }
    r.l[13] = (u32)(acc + (int64_t)a.l[13] - (int64_t)((u64)q * fqu_p(13)));
    return r;
}
// a - b - c + 8 p, normalised: a, b, c normalised, b + c < 8 p
__device__ __forceinline__ FqU fqu_sub2_norm(const FqU& a, const FqU& b, const FqU& c) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fqu_8p_wide(i) - b.l[i] - c.l[i]);
    return fqu_normalize(r);
}
__device__ __forceinline__ FqU fqu_r3() {   // R'^3 mod p: fqu_mul(integer, R'^3) = integer * R'^2
    constexpr u32 m[14] = {0xf63e3ebu, 0xd055de1u, 0x6ff6650u, 0xd6bd950u, 0x9cd510eu, 0x09ed341u, 0x11a3aa6u,
                           0x40b6ca4u, 0x200fa40u, 0x28c4a35u, 0x8a2198cu, 0x956bce5u, 0x96dd52au, 0x5ffu};
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = m[i];
    return r;
}
struct AffWaveDone {   // stop the division steps once every lane of the wave has g == 0
    __device__ bool operator()(bool mine) const { return __all(mine); }
};
// inverse of a U-form value a R' (normalised limbs, < 2 p): a^-1 R' (multiply output)
__device__ __forceinline__ FqU fqu_inv(const FqU& a) {
    Fq w = fqu_pack(a);
    fp_reduce(w);
    Fq i = fq_inv_safegcd_words(w, AffWaveDone{});
    return fqu_mul(fqu_unpack(i), fqu_r3());
}

// ---- record builders (one thread per bucket) ------------------------------------------------------------------------------
// rank of bucket b in the memory order of `sorted` (partitioned sort: partition = low bits; one-pass sort: n_parts == 0)
__device__ __forceinline__ u32 aff_rank(u32 b, unsigned n_parts, unsigned part_shift, unsigned part_log) {
    return n_parts ? (((b & (n_parts - 1u)) << part_log) | (b >> part_shift)) : b;
}
// round 1: sources are the sorted table codes of the bucket (at most cap of them: the rest is the over-full path's)
__global__ void k_aff_build_first(const u32* sorted, size_t sorted_stride, const u32* offsets, const u32* counts, size_t B, unsigned n_parts,
                                  unsigned part_shift, unsigned part_log, u32 cap, size_t S1, uint2* rec, u32* off1, u32* cnt1) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned lane = blockIdx.y;
    const u32* srt = sorted + (size_t)lane * sorted_stride;
    const u32 off0 = offsets[lane * B + b];
    u32 cnt = counts[lane * B + b];
    if (cnt > cap) cnt = cap;
    const u32 o1 = (off0 + aff_rank((u32)b, n_parts, part_shift, part_log) + 1u) >> 1;
    off1[lane * B + b] = (u32)(lane * S1) + o1;            // absolute slot index
    cnt1[lane * B + b] = (cnt + 1u) >> 1;
    uint2* r = rec + lane * S1 + o1;
    for (u32 i = 0; i < cnt / 2; i++) r[i] = make_uint2(srt[off0 + 2 * i], srt[off0 + 2 * i + 1]);
    if (cnt & 1u) r[cnt / 2] = make_uint2(srt[off0 + cnt - 1], AFF_NONE);
}
// round r >= 2: sources are slots of level r - 1 (absolute indices)
__global__ void k_aff_build_next(const u32* off_in, const u32* cnt_in, size_t B, unsigned n_parts, unsigned part_shift, unsigned part_log, size_t S_in,
                                 size_t S_out, uint2* rec, u32* off_out, u32* cnt_out) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned lane = blockIdx.y;
    const u32 oi = off_in[lane * B + b], cnt = cnt_in[lane * B + b];
    const u32 local = oi - (u32)(lane * S_in);
    const u32 oo = (local + aff_rank((u32)b, n_parts, part_shift, part_log) + 1u) >> 1;
    off_out[lane * B + b] = (u32)(lane * S_out) + oo;
    cnt_out[lane * B + b] = (cnt + 1u) >> 1;
    uint2* r = rec + lane * S_out + oo;
    for (u32 i = 0; i < cnt / 2; i++) r[i] = make_uint2(oi + 2 * i, oi + 2 * i + 1);
    if (cnt & 1u) r[cnt / 2] = make_uint2(oi + cnt - 1, AFF_NONE);
}

// ---- the round kernel -----------------------------------------------------------------------------------------------------
template <bool FROM_TABLE>
__device__ __forceinline__ void aff_load_x(const u64* pts, const uint4* src, u32 id, FqU& x, u32& flags) {
    if constexpr (FROM_TABLE) {
        table_load_x(pts, id, x);
        flags = 0;
    } else {
        lvl_load_coord(src, id, 0, x, &flags);
    }
}
template <bool FROM_TABLE>
__device__ __forceinline__ void aff_load_y(const u64* pts, const uint4* src, u32 id, FqU& y) {
    if constexpr (FROM_TABLE) table_load_y(pts, id, y);
    else lvl_load_coord(src, id, 4, y, nullptr);
}
// x2 - x1 + 4 p (lazy) and whether the pair needs the complete formulas: an infinity operand, or x2 == x1 mod p possible
// (d = j p for some j in 1..8 -- only then can d vanish mod p -- and p == 1 mod 2^28 makes the low limb of j p equal j)
__device__ __forceinline__ bool aff_delta(const FqU& x1, const FqU& x2, u32 f1, u32 f2, FqU& d) {
    d = fqu_sub_lazy<4>(x2, x1);
    return ((f1 | f2) & AFF_F_INF) || (((d.l[0] & FQU_MASK) - 1u) <= 7u);
}

__device__ __forceinline__ size_t pre_idx(unsigned wave, unsigned j, int c, unsigned lane) { return ((((size_t)wave * AFF_KB + j) * 4 + c) << 6) | lane; }

template <bool FROM_TABLE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_affine_round(const uint2* rec, size_t total, const u64* pts, const uint4* src,
                                                                                               uint4* dst, uint8_t* pend, uint4* scratch) {
    const unsigned lane = threadIdx.x, wave = blockIdx.x;
    for (size_t item = wave; item * AFF_ITEM < total; item += gridDim.x) {
        const size_t base = item * AFF_ITEM;
        // forward: running products of the denominators
        FqU run = fqu_one();
        for (unsigned j = 0; j < AFF_KB; j++) {
            const size_t s = base + (size_t)j * 64 + lane;
            uint2 r = s < total ? rec[s] : make_uint2(AFF_NONE, AFF_NONE);
            if (r.x != AFF_NONE && r.y != AFF_NONE) {
                FqU x1, x2, d;
                u32 f1, f2;
                aff_load_x<FROM_TABLE>(pts, src, r.x, x1, f1);
                aff_load_x<FROM_TABLE>(pts, src, r.y, x2, f2);
                if (!aff_delta(x1, x2, f1, f2, d)) {
                    const Fq pk = fqu_pack(run);
                    scratch[pre_idx(wave, j, 0, lane)] = make_uint4(pk.l[0], pk.l[1], pk.l[2], pk.l[3]);
                    scratch[pre_idx(wave, j, 1, lane)] = make_uint4(pk.l[4], pk.l[5], pk.l[6], pk.l[7]);
                    scratch[pre_idx(wave, j, 2, lane)] = make_uint4(pk.l[8], pk.l[9], pk.l[10], pk.l[11]);
                    run = fqu_mul(run, d);
                }
            }
        }
        FqU inv = fqu_inv(run);
        // backward: unwind the products, finish the additions
        for (int j = (int)AFF_KB - 1; j >= 0; j--) {
            const size_t s = base + (size_t)j * 64 + lane;
            uint2 r = s < total ? rec[s] : make_uint2(AFF_NONE, AFF_NONE);
            if (r.x == AFF_NONE) continue;
            if (r.y == AFF_NONE) {                       // a bucket's odd point out: carried over unchanged
                AffPoint p;
                aff_load_x<FROM_TABLE>(pts, src, r.x, p.x, p.flags);
                aff_load_y<FROM_TABLE>(pts, src, r.x, p.y);
                lvl_store(dst, s, p);
                continue;
            }
            FqU x1, x2, d;
            u32 f1, f2;
            aff_load_x<FROM_TABLE>(pts, src, r.x, x1, f1);
            aff_load_x<FROM_TABLE>(pts, src, r.y, x2, f2);
            if (aff_delta(x1, x2, f1, f2, d)) {
                pend[s] = 1;                             // k_affine_fix recomputes this slot with the complete formulas
                continue;
            }
            FqU y1, y2;
            aff_load_y<FROM_TABLE>(pts, src, r.x, y1);
            aff_load_y<FROM_TABLE>(pts, src, r.y, y2);
            Fq pk;
            {
                uint4 a = scratch[pre_idx(wave, j, 0, lane)], b = scratch[pre_idx(wave, j, 1, lane)], c = scratch[pre_idx(wave, j, 2, lane)];
                pk.l[0] = a.x; pk.l[1] = a.y; pk.l[2] = a.z; pk.l[3] = a.w;
                pk.l[4] = b.x; pk.l[5] = b.y; pk.l[6] = b.z; pk.l[7] = b.w;
                pk.l[8] = c.x; pk.l[9] = c.y; pk.l[10] = c.z; pk.l[11] = c.w;
            }
            const FqU dinv = fqu_mul(inv, fqu_unpack(pk));                               // 1 / (x2 - x1)
            inv = fqu_mul(inv, d);
            const FqU lam = fqu_mul(fqu_sub_lazy<8>(y2, y1), dinv);                      // y1 <= 4 p (a negated table point)
            AffPoint o;
            o.x = fqu_reduce_small(fqu_sub2_norm(fqu_sqr(lam), x1, x2));                 // lambda^2 - x1 - x2 (+ 8 p), then < 3 p
            o.y = fqu_reduce_small(fqu_normalize(fqu_sub_lazy<8>(fqu_mul(lam, fqu_sub_lazy<4>(x1, o.x)), y1)));
            o.flags = 0;
            lvl_store(dst, s, o);
        }
    }
}

// ---- complete formulas for the flagged slots ---------------------------------------------------------------------------------
// saturated Montgomery coordinate from an unsaturated one (value < 2^384) and back
__device__ __forceinline__ Fq aff_to_sat(const FqU& a) { return fp_mul(fqu_pack(fqu_normalize(a)), fqu_k_from_u()); }
__device__ __forceinline__ FqU aff_from_sat(const Fq& a) { return fqu_unpack(fp_mul(a, fqu_k_to_u())); }

template <bool FROM_TABLE>
__global__ __launch_bounds__(256) void k_affine_fix(const uint2* rec, size_t total, const u64* pts, const uint4* src, uint4* dst, const uint8_t* pend) {
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total || !pend[s]) return;
    const uint2 r = rec[s];
    AffPoint a, b;
    aff_load_x<FROM_TABLE>(pts, src, r.x, a.x, a.flags);
    aff_load_y<FROM_TABLE>(pts, src, r.x, a.y);
    aff_load_x<FROM_TABLE>(pts, src, r.y, b.x, b.flags);
    aff_load_y<FROM_TABLE>(pts, src, r.y, b.y);
    // short_weierstrass_jacobian.rs:570-597: infinity operands, equal points (-> doubling), opposite points (-> infinity)
    Jac<Fq> acc = (a.flags & AFF_F_INF) ? Jac<Fq>::zero() : Jac<Fq>{aff_to_sat(a.x), aff_to_sat(a.y), Fq::one()};
    acc = jac_add_mixed(acc, Affine<Fq>{aff_to_sat(b.x), aff_to_sat(b.y)}, (b.flags & AFF_F_INF) != 0);
    Affine<Fq> res;
    AffPoint o;
    if (jac_to_affine(acc, res)) {
        o.x = fqu_one();
        o.y = fqu_one();
        o.flags = AFF_F_INF;
    } else {
        o.x = aff_from_sat(res.x);
        o.y = aff_from_sat(res.y);
        o.flags = 0;
    }
    lvl_store(dst, s, o);
}

// ---- XYZZ accumulation of what the rounds left (the analogue of k_accumulate_u, sources = level slots) ------------------------
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_accumulate_u_lvl(
    const uint4* lvl, const u32* off_r, const u32* cnt_r, const u32* perm, size_t B, u64* buckets, uint8_t* dirty, u32* exc_count, u32* exc_list, u32 exc_cap) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const unsigned lane = blockIdx.y;
    const size_t b = perm[(size_t)lane * B + t];
    const u32 off = off_r[lane * B + b], cnt = cnt_r[lane * B + b];
    FqU ax, ay, azz, azzz;
    bool inf = true;
    for (u32 e = 0; e < cnt; e++) {
        AffPoint q = lvl_load(lvl, off + e);
        if (q.flags & AFF_F_INF) continue;
        if (inf) {
            ax = q.x;
            ay = q.y;
            azz = fqu_one();
            azzz = azz;
            inf = false;
            continue;
        }
        if (!fqu_xyzz_acc_mixed(ax, ay, azz, azzz, q.x, q.y)) {
            u32 slot = atomicAdd(exc_count, 1u);
            if (slot < exc_cap) {
                exc_list[3 * slot] = lane;
                exc_list[3 * slot + 1] = (u32)b;
                exc_list[3 * slot + 2] = off + e;
                continue;
            }
            dirty[(size_t)lane * B + b] = 1;
            return;
        }
    }
    XYZZ<Fq> out = XYZZ<Fq>::zero();
    if (!inf) {
        const Fq kf = fqu_k_from_u();
        out.x = fp_mul(fqu_pack(ax), kf);
        out.y = fp_mul(fqu_pack(ay), kf);
        out.zz = fp_mul(fqu_pack(azz), kf);
        out.zzz = fp_mul(fqu_pack(azzz), kf);
    }
    xyzz_store<Fq>(buckets + (size_t)24 * ((size_t)lane * B + b), out);
}
// adds the deferred level points into the finished buckets (buckets recomputed from scratch by k_accumulate_u_fix already hold theirs)
__global__ void k_accumulate_u_lvl_cleanup(const uint4* lvl, size_t B, u64* buckets, const uint8_t* dirty, const u32* exc_count, const u32* exc_list, u32 exc_cap) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    u32 n = *exc_count;
    if (n > exc_cap) n = exc_cap;
    for (u32 k = 0; k < n; k++) {
        u32 lane = exc_list[3 * k], b = exc_list[3 * k + 1], slot = exc_list[3 * k + 2];
        if (dirty[(size_t)lane * B + b]) continue;
        u64* bs = buckets + (size_t)24 * ((size_t)lane * B + b);
        XYZZ<Fq> acc = xyzz_load<Fq>(bs);
        AffPoint q = lvl_load(lvl, slot);
        if (q.flags & AFF_F_INF) continue;
        xyzz_acc_mixed(acc.x, acc.y, acc.zz, acc.zzz, aff_to_sat(q.x), aff_to_sat(q.y));
        xyzz_store<Fq>(bs, acc);
    }
}

}  // namespace czk

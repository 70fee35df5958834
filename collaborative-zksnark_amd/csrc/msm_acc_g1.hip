// msm_acc_g1.hip -- bucket accumulation kernel instantiated for G1 (base field Fq).
#define CZK_FQU_G1 1
#include "fqu.h"
#include "te.h"
#include "msm_acc.h"
#ifdef CZK_LAB
#include "lab/msm_aff.h"
#endif

namespace czk {
#ifdef CZK_LAB   // saturated tables (lab option "msm_sat"): the round-1 kernel
void launch_accumulate_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes) {
    hipLaunchKernelGGL(k_accumulate<Fq>, dim3((unsigned)((B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets);
}
#endif
// The unsaturated accumulation is issued in three pieces so that only the dominant kernel occupies the accumulate
// stream: `prepare` (clear the dirty flags + exception list; sort stream), the kernel itself, `fixup` (recompute dirty
// buckets, add deferred points; reduce stream, ahead of the bucket reduction).
static constexpr u32 G1_EXC_CAP = 4096;
void launch_accumulate_g1_u_prepare(hipStream_t st, uint8_t* dirty, size_t B, unsigned lanes) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    (void)hipMemsetAsync(dirty, 0, flags + 16, st);
}
void launch_accumulate_g1_u(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                            size_t sorted_stride, u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets) {
    // workspace behind the dirty flags: [count][list of EXC_CAP x 3 u32]
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    ProfScope ps(ctx, "msm_accumulate_g1", st);   // brackets the dominant kernel only
    hipLaunchKernelGGL(k_accumulate_u, dim3((unsigned)((B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets, dirty, exc, exc + 4, G1_EXC_CAP, ubuckets);
}
void launch_accumulate_g1_u_fixup(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                  u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    hipLaunchKernelGGL(k_accumulate_u_fix, dim3(fix_grid(B), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, B,
                       sorted_stride, buckets, dirty, ubuckets);
    hipLaunchKernelGGL(k_accumulate_u_cleanup, dim3(1), dim3(64), 0, st, pts, B, buckets, dirty, exc, exc + 4, G1_EXC_CAP, ubuckets);
}
#ifdef CZK_LAB
// ---- batched-affine pre-reduction (lab/msm_aff.h; measured slower, EXPERIMENTS.md) --------------------------------------------
// Slot counts of the levels: S_r = round_up_64((S_{r-1} + B + 1) / 2 + 1) with S_0 = entries per lane.
void aff_plan(size_t total0, size_t B, unsigned rounds, size_t* S) {
    size_t prev = total0;
    for (unsigned r = 0; r < rounds; r++) {
        S[r] = (((prev + B + 1) / 2 + 1) + 63) & ~(size_t)63;
        prev = S[r];
    }
}
// all records are a function of (offsets, counts) alone: built up front on the sort stream
void launch_affine_build_g1(hipStream_t st, const AffArgs& a) {
    const dim3 grid((unsigned)((a.B + 255) / 256), a.lanes), block(256);
    for (unsigned r = 0; r < a.rounds; r++) {
        (void)hipMemsetAsync(a.rec[r], 0xff, (size_t)a.lanes * a.S[r] * sizeof(uint2), st);
        (void)hipMemsetAsync(a.pend[r], 0, (size_t)a.lanes * a.S[r], st);
        if (r == 0)
            hipLaunchKernelGGL(k_aff_build_first, grid, block, 0, st, a.sorted, a.sorted_stride, a.offsets, a.counts, a.B, a.n_parts, a.part_shift, a.part_log,
                               HEAVY_CHUNK, a.S[0], (uint2*)a.rec[0], a.off[0], a.cnt[0]);
        else
            hipLaunchKernelGGL(k_aff_build_next, grid, block, 0, st, a.off[(r - 1) & 1], a.cnt[(r - 1) & 1], a.B, a.n_parts, a.part_shift, a.part_log, a.S[r - 1],
                               a.S[r], (uint2*)a.rec[r], a.off[r & 1], a.cnt[r & 1]);
    }
}
// the rounds, then the XYZZ accumulation of the last level; `st` = the accumulate stream
void launch_affine_accumulate_g1(czk_ctx* ctx, hipStream_t st, const AffArgs& a, const u64* pts, const u32* perm, u64* buckets, uint8_t* dirty) {
    size_t flags = ((size_t)a.lanes * a.B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    ProfScope ps(ctx, "msm_accumulate_g1", st);
    const unsigned waves = (unsigned)ctx->num_cu * 8;   // two resident waves per SIMD, each looping over work items
    for (unsigned r = 0; r < a.rounds; r++) {
        const size_t total = (size_t)a.lanes * a.S[r];
        const uint4* src = r == 0 ? nullptr : (const uint4*)a.lvl[(r - 1) & 1];
        uint4* dst = (uint4*)a.lvl[r & 1];
        if (r == 0) {
            hipLaunchKernelGGL(k_affine_round<true>, dim3(waves), dim3(64), 0, st, (const uint2*)a.rec[r], total, pts, src, dst, a.pend[r], (uint4*)a.scratch);
            hipLaunchKernelGGL(k_affine_fix<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint2*)a.rec[r], total, pts, src, dst, a.pend[r]);
        } else {
            hipLaunchKernelGGL(k_affine_round<false>, dim3(waves), dim3(64), 0, st, (const uint2*)a.rec[r], total, pts, src, dst, a.pend[r], (uint4*)a.scratch);
            hipLaunchKernelGGL(k_affine_fix<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint2*)a.rec[r], total, pts, src, dst, a.pend[r]);
        }
    }
    const unsigned last = a.rounds - 1;
    hipLaunchKernelGGL(k_accumulate_u_lvl, dim3((unsigned)((a.B + 127) / 128), a.lanes), dim3(128), 0, st, (const uint4*)a.lvl[last & 1], a.off[last & 1], a.cnt[last & 1], perm,
                       a.B, buckets, dirty, exc, exc + 4, G1_EXC_CAP);
}
size_t aff_scratch_bytes(czk_ctx* ctx) { return (size_t)ctx->num_cu * 8 * AFF_KB * 4 * 64 * 16; }
void launch_accumulate_g1_u_fixup_lvl(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                      u64* buckets, unsigned lanes, uint8_t* dirty, const void* lvl) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    hipLaunchKernelGGL(k_accumulate_u_fix, dim3(fix_grid(B), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, B,
                       sorted_stride, buckets, dirty, 0);
    hipLaunchKernelGGL(k_accumulate_u_lvl_cleanup, dim3(1), dim3(64), 0, st, (const uint4*)lvl, B, buckets, dirty, exc, exc + 4, G1_EXC_CAP);
}
#endif   // CZK_LAB
void launch_convert_to_u(hipStream_t st, u64* pts, size_t n_coords) {
    hipLaunchKernelGGL(k_convert_to_u, dim3((unsigned)((n_coords + 255) / 256)), dim3(256), 0, st, pts, n_coords);
}
void launch_convert_from_u(hipStream_t st, u64* pts, size_t n_coords) {
    hipLaunchKernelGGL(k_convert_from_u, dim3((unsigned)((n_coords + 255) / 256)), dim3(256), 0, st, pts, n_coords);
}
#ifdef CZK_LAB   // bucket reduction in the saturated form (lab option "msm_reduce_sat")
void launch_reduce_level_g1(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes) {
    hipLaunchKernelGGL((k_reduce_level<Fq, 24, 0>), dim3((unsigned)(((n_out << 0) + 127) / 128), lanes), dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
}
void launch_finish_g1(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out) {
    hipLaunchKernelGGL((k_finish<Fq, 24, 0>), dim3((unsigned)(((segs << 0) + 63) / 64)), dim3(64), 0, st, P, E, segs, out);
}
void launch_reduce_tail_g1(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out,
                           unsigned lanes) {
    hipLaunchKernelGGL((k_reduce_tail_sums<Fq, 24, 0>), dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
    hipLaunchKernelGGL((k_reduce_tail_finish<Fq, 24, 0>), dim3((lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
}
#endif   // CZK_LAB
// the three steps of the bucket reduction on u-form buckets (msm_acc.h k_reduce_*_p); te = twisted Edwards buckets (te.h)
void launch_reduce_level_g1_u(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                              unsigned lanes, int te) {
    const dim3 grid((unsigned)((n_out + 127) / 128), lanes);
    if (te) hipLaunchKernelGGL(k_reduce_level_p<TeOps>, grid, dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
    else hipLaunchKernelGGL(k_reduce_level_p<XyzzOps>, grid, dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
}
void launch_finish_g1_u(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out, int te) {
    const dim3 grid((unsigned)((segs + 63) / 64));
    if (te) hipLaunchKernelGGL(k_finish_p<TeOps>, grid, dim3(64), 0, st, P, E, segs, out);
    else hipLaunchKernelGGL(k_finish_p<XyzzOps>, grid, dim3(64), 0, st, P, E, segs, out);
}
void launch_reduce_tail_g1_u(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out, unsigned lanes,
                             int te) {
    if (te) {
        hipLaunchKernelGGL(k_reduce_tail_sums_p<TeOps>, dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
        hipLaunchKernelGGL(k_reduce_tail_finish_p<TeOps>, dim3((lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
    } else {
        hipLaunchKernelGGL(k_reduce_tail_sums_p<XyzzOps>, dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
        hipLaunchKernelGGL(k_reduce_tail_finish_p<XyzzOps>, dim3((lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
    }
}
// ---- twisted Edwards form (te.h) ----------------------------------------------------------------------------------------------------
// table conversion at registration: n SW affine Montgomery points -> n x 24 u64 niels entries (te.h layout); *bad (device u32) is set when a
// point has no image
void launch_sw_to_te_niels(hipStream_t st, const u64* aff, const uint8_t* inf, size_t n, u64* scratch, u64* out, u32* bad) {
    const unsigned CH = 32;
    hipLaunchKernelGGL(k_sw_to_te_niels, dim3((unsigned)(((n + CH - 1) / CH + 127) / 128)), dim3(128), 0, st, aff, inf, n, CH, scratch, out, bad);
}
void launch_accumulate_g1_te(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                             size_t sorted_stride, u64* buckets, unsigned lanes) {
    ProfScope ps(ctx, "msm_accumulate_g1", st);   // brackets the dominant kernel only
    // (a build for 3 waves per SIMD -- the kernel needs 155 VGPRs -- was measured: same isolated time, 88.6 against 84.0 ms per proof)
    const unsigned G = acc_interleave(ctx, lanes);
    hipLaunchKernelGGL(k_accumulate_te, dim3((unsigned)((B * G + 127) / 128), (lanes + G - 1) / G), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B, sorted_stride,
                       buckets, G, lanes);
}
void launch_heavy_g1_te(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride, u64* buckets,
                        unsigned lanes, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap) {
    (void)hipMemsetAsync(hdr, 0, 16, st);
    hipLaunchKernelGGL(k_heavy_list<Fq>, dim3((unsigned)((B + 255) / 256), lanes), dim3(256), 0, st, counts, B, hdr, items, heavy, cap);
    hipLaunchKernelGGL(k_accumulate_heavy_te, dim3((cap + 127) / 128), dim3(128), 0, st, pts, sorted, offsets, counts, B, sorted_stride, hdr, items, partials, cap);
    hipLaunchKernelGGL(k_heavy_combine_te, dim3(cap / 4 + 1), dim3(128), 0, st, hdr, heavy, partials, B, buckets, cap);
}
// over-full buckets (see msm_acc.h): item list, per-item partial sums, combination into the buckets; all on `st`
void launch_heavy_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                     u64* buckets, unsigned lanes, const uint8_t* dirty, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap, int unsat, int ubuckets) {
    (void)hipMemsetAsync(hdr, 0, 16, st);
    hipLaunchKernelGGL(k_heavy_list<Fq>, dim3((unsigned)((B + 255) / 256), lanes), dim3(256), 0, st, counts, B, hdr, items, heavy, cap);
    hipLaunchKernelGGL(k_accumulate_heavy<Fq>, dim3((cap + 127) / 128), dim3(128), 0, st, pts, sorted, offsets, counts, B, sorted_stride, hdr, items,
                       partials, cap, unsat);
    hipLaunchKernelGGL(k_heavy_combine<Fq>, dim3(cap / 4 + 1), dim3(128), 0, st, hdr, heavy, partials, B, buckets, dirty, cap, ubuckets);   // a bucket is over-full above 1024 entries: at most total / 1024 of them
}
}  // namespace czk

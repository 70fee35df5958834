// msm_acc_g2.hip -- bucket accumulation kernel instantiated for G2 (base field Fq2).
#include <stdlib.h>

#define CZK_FQU_G2 1
#include "fq2p.h"
#include "fqu.h"
#ifdef CZK_LAB   // the lane-pair accumulate kernel k_accumulate_u2p (faster alone, slower per proof) and the interleaved-chain products
#define CZK_FQ2PU 1
#include "lab/fqu_il.h"
#include "fq2pu.h"
#endif
#include "msm_acc.h"

namespace czk {
#ifdef CZK_LAB
void launch_accumulate_g2(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes) {
    // saturated tables (lab option "msm_sat_g2" at registration): one bucket per thread
    hipLaunchKernelGGL(k_accumulate<Fq2>, dim3((unsigned)((B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets);
}
#endif
static constexpr u32 G2_EXC_CAP = 4096;
void launch_accumulate_g2_u_prepare(hipStream_t st, uint8_t* dirty, size_t B, unsigned lanes) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    (void)hipMemsetAsync(dirty, 0, flags + 16, st);
}
void launch_accumulate_g2_u(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                            size_t sorted_stride, u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    ProfScope ps(ctx, "msm_accumulate_g2", st);   // brackets the dominant kernel only
#ifdef CZK_LAB
    if (ubuckets && ctx->msm_g2_mode == 1)
        hipLaunchKernelGGL((k_accumulate_u2p<128, 2>), dim3((unsigned)((2 * B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B, sorted_stride,
                           buckets, dirty, exc, exc + 4, G2_EXC_CAP);
    else if (ubuckets && ctx->msm_g2_mode == 2)
        hipLaunchKernelGGL((k_accumulate_u2p<512, 3>), dim3((unsigned)((2 * B + 511) / 512), lanes), dim3(512), 82 * 1024, st, pts, sorted, offsets, counts, perm, B,
                           sorted_stride, buckets, dirty, exc, exc + 4, G2_EXC_CAP);
    else
#endif
    {
        const unsigned G = acc_interleave(ctx, lanes);
        hipLaunchKernelGGL(k_accumulate_u2, dim3((unsigned)((B * G + 127) / 128), (lanes + G - 1) / G), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                           sorted_stride, buckets, dirty, exc, exc + 4, G2_EXC_CAP, ubuckets, G, lanes);
    }
}
void launch_accumulate_g2_u_fixup(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                  u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets) {
    size_t flags = ((size_t)lanes * B + 15) & ~(size_t)15;
    u32* exc = (u32*)(dirty + flags);
    hipLaunchKernelGGL(k_accumulate_u2_fix, dim3(fix_grid(B), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, B,
                       sorted_stride, buckets, dirty, ubuckets);
    hipLaunchKernelGGL(k_accumulate_u2_cleanup, dim3(1), dim3(64), 0, st, pts, B, buckets, dirty, exc, exc + 4, G2_EXC_CAP, ubuckets);
}
}  // namespace czk

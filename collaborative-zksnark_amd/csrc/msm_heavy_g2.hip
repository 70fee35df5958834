// msm_heavy_g2.hip -- the over-full-bucket kernels for G2 (see msm_acc.h); split out for build time.
#include "fq2p.h"
#include "fqu.h"
#include "msm_acc.h"

namespace czk {
// over-full buckets (see msm_acc.h): item list, per-item partial sums, combination into the buckets; all on `st`
void launch_heavy_g2(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                     u64* buckets, unsigned lanes, const uint8_t* dirty, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap, int unsat, int ubuckets) {
    (void)hipMemsetAsync(hdr, 0, 16, st);
    hipLaunchKernelGGL(k_heavy_list<Fq2>, dim3((unsigned)((B + 255) / 256), lanes), dim3(256), 0, st, counts, B, hdr, items, heavy, cap);
    hipLaunchKernelGGL(k_accumulate_heavy<Fq2>, dim3((cap + 127) / 128), dim3(128), 0, st, pts, sorted, offsets, counts, B, sorted_stride, hdr, items,
                       partials, cap, unsat);
    hipLaunchKernelGGL(k_heavy_combine<Fq2>, dim3(cap / 4 + 1), dim3(128), 0, st, hdr, heavy, partials, B, buckets, dirty, cap, ubuckets);   // a bucket is over-full above 1024 entries: at most total / 1024 of them
}
}  // namespace czk

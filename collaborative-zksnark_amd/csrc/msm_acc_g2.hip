// msm_acc_g2.hip -- bucket accumulation kernel instantiated for G2 (base field Fq2).
#include "fq2p.h"
#include "msm_acc.h"

namespace czk {
void launch_accumulate_g2(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes) {
    // one bucket per lane pair: 64 buckets per 128-thread workgroup
    hipLaunchKernelGGL(k_accumulate_pair<Fq2P>, dim3((unsigned)((B + 63) / 64), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets);
}
}  // namespace czk

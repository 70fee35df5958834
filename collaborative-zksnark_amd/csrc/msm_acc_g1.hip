// msm_acc_g1.hip -- bucket accumulation kernel instantiated for G1 (base field Fq).
#include "msm_acc.h"

namespace czk {
void launch_accumulate_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes) {
    hipLaunchKernelGGL(k_accumulate<Fq>, dim3((unsigned)((B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets);
}
}  // namespace czk

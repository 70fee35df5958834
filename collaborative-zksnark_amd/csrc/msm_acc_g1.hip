// msm_acc_g1.hip -- bucket accumulation kernel instantiated for G1 (base field Fq).
#include "msm_acc.h"

namespace czk {
void launch_accumulate_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes) {
    hipLaunchKernelGGL(k_accumulate<Fq>, dim3((unsigned)((B + 127) / 128), lanes), dim3(128), 0, st, pts, sorted, offsets, counts, perm, B,
                       sorted_stride, buckets);
}
void launch_reduce_level_g1(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes) {
    hipLaunchKernelGGL((k_reduce_level<Fq, 24, 0>), dim3((unsigned)(((n_out << 0) + 127) / 128), lanes), dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
}
void launch_finish_g1(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out) {
    hipLaunchKernelGGL((k_finish<Fq, 24, 0>), dim3((unsigned)(((segs << 0) + 63) / 64)), dim3(64), 0, st, P, E, segs, out);
}
}  // namespace czk

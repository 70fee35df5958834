// msm_red_g2.hip -- G2 bucket reduction (lane-pair Fq2 arithmetic, fq2p.h) and the over-full-bucket kernels for G2;
// a translation unit of its own because these instantiations dominate the build time.
#include "fq2p.h"
#include "fqu.h"
#include "msm_acc.h"

namespace czk {
void launch_reduce_level_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes) {
    hipLaunchKernelGGL((k_reduce_level<Fq2P, 48, 1>), dim3((unsigned)(((n_out << 1) + 127) / 128), lanes), dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
}
void launch_reduce_tail_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out,
                           unsigned lanes) {
    hipLaunchKernelGGL((k_reduce_tail_sums<Fq2P, 48, 1>), dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
    hipLaunchKernelGGL((k_reduce_tail_finish<Fq2P, 48, 1>), dim3((2 * lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
}
void launch_finish_g2(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out) {
    hipLaunchKernelGGL((k_finish<Fq2P, 48, 1>), dim3((unsigned)(((segs << 1) + 63) / 64)), dim3(64), 0, st, P, E, segs, out);
}
}  // namespace czk

// msm_red_g2.hip -- G2 bucket reduction: on unsaturated lane pairs (fq2pu.h, buckets in u-form; the product path) or, in the lab build only,
// on saturated lane pairs (fq2p.h; options "msm_reduce_sat", "msm_reduce_sat_g2", and handles registered under "msm_sat_g2"); a translation unit of its own because these instantiations
// dominate the build time.
#include "fq2p.h"
#include "fqu.h"
#include "fq2pu.h"
#include "msm_acc.h"

#ifndef CZK_G2RED_WAVES
#define CZK_G2RED_WAVES 1   // waves per SIMD of the u-form reduction kernels: at 2 (256 VGPRs) k_reduce_level_p spills 111 registers; same-box A/B 78.2 (1) / 79.9 (2) / 80.6 (3) ms per proof
#endif

namespace czk {
struct Xyzz2Ops {
    static constexpr int WAVES = CZK_G2RED_WAVES, JW = 48, JACW = 36, SHIFT = 1;
    typedef XYZZU2 P;
    static __device__ __forceinline__ P zero() { return xyzzu2_zero(); }
    static __device__ __forceinline__ P load(const u64* p) { return xyzzu2_load(p); }
    static __device__ __forceinline__ void store(u64* p, const P& a) { xyzzu2_store(p, a); }
    static __device__ __forceinline__ void add(P& a, const P& b) { xyzzu2_add(a, b); }
    static __device__ __forceinline__ void dbl(P& a) { xyzzu2_double(a); }
    static __device__ __forceinline__ void store_jac(u64* out, const P& a) { jac_store<Fq2P>(out, xyzz_to_jac(xyzzu2_to_sat(a))); }
};
void launch_reduce_level_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes, int ub) {
    const dim3 grid((unsigned)(((n_out << 1) + 127) / 128), lanes);
    if (ub) hipLaunchKernelGGL(k_reduce_level_p<Xyzz2Ops>, grid, dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
#ifdef CZK_LAB
    else hipLaunchKernelGGL((k_reduce_level<Fq2P, 48, 1>), grid, dim3(128), 0, st, P, E, n_in, L, scale_dbl, Po, Eo, n_out);
#endif
}
void launch_reduce_tail_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out,
                           unsigned lanes, int ub) {
    if (ub) {
        hipLaunchKernelGGL(k_reduce_tail_sums_p<Xyzz2Ops>, dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
        hipLaunchKernelGGL(k_reduce_tail_finish_p<Xyzz2Ops>, dim3((2 * lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
        return;
    }
#ifdef CZK_LAB
    hipLaunchKernelGGL((k_reduce_tail_sums<Fq2P, 48, 1>), dim3(TAIL_BLOCKS, lanes), dim3(TAIL_THREADS), 0, st, P, E, n_in, scratch, sums);
    hipLaunchKernelGGL((k_reduce_tail_finish<Fq2P, 48, 1>), dim3((2 * lanes + 63) / 64), dim3(64), 0, st, sums, scale_dbl, (size_t)lanes, out);
#endif
}
void launch_finish_g2(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out, int ub) {
    const dim3 grid((unsigned)(((segs << 1) + 63) / 64));
    if (ub) hipLaunchKernelGGL(k_finish_p<Xyzz2Ops>, grid, dim3(64), 0, st, P, E, segs, out);
#ifdef CZK_LAB
    else hipLaunchKernelGGL((k_finish<Fq2P, 48, 1>), grid, dim3(64), 0, st, P, E, segs, out);
#endif
}
}  // namespace czk

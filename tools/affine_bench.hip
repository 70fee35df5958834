// affine_bench.hip -- prototype + measurement of BATCHED-AFFINE bucket accumulation for G1 (developer tool, not product).
//
// Question: is an affine addition with a shared inversion cheaper on gfx950 than the XYZZ mixed
// addition the accumulate kernel uses (4 490 VALU instructions, 3 416 of them v_mad_u64_u32)?
//   affine:  lambda = (y2 - y1) / (x2 - x1);  x3 = lambda^2 - x1 - x2;  y3 = lambda (x1 - x3) - y1
//   with Montgomery's trick over a batch of K independent additions:  5 M + 1 S per addition + one inversion per batch.
// Every lane of a wave executes the inversion whether or not other lanes need one, so the batch must be K additions of
// the SAME thread; the inversion is Bernstein-Yang safegcd (csrc/lab/fq_safegcd.h, ~25 k instructions) instead of Fermat
// (~270 k).  Each thread owns K independent (P1_j, P2_j) pairs -- the shape of "thread owns K buckets and adds the next
// point to each" -- kept in global memory in 16-byte-interleaved SoA order (coalesced), prefix products in a global scratch.
// The kernel iterates P1_j <- P1_j + P2_j; results are dumped for an independent big-integer check (tools/affine_check.py).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../collaborative-zksnark_amd/csrc affine_bench.hip -o affine_bench.bin
// Run:   ./affine_bench.bin [dump.bin]       (rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace for instruction counts)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "curve.h"
#include "fqu.h"
#include "lab/fq_safegcd.h"
using namespace czk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// element (slot j, chunk c of NC, thread t) of an interleaved array of NC x 16-byte chunks per element
template <int NC>
__device__ __forceinline__ size_t il(size_t j, int c, size_t T, size_t t) { return ((j * NC + c) * T + t); }

__device__ __forceinline__ Fq ld_fq(const uint4* base, size_t j, int c0, size_t T, size_t t) {   // one coordinate = 3 chunks
    Fq r;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint4 v = base[il<6>(j, c0 + c, T, t)];
        r.l[4 * c] = v.x; r.l[4 * c + 1] = v.y; r.l[4 * c + 2] = v.z; r.l[4 * c + 3] = v.w;
    }
    return r;
}
__device__ __forceinline__ void st_fq(uint4* base, size_t j, int c0, size_t T, size_t t, const Fq& a) {
#pragma unroll
    for (int c = 0; c < 3; c++) base[il<6>(j, c0 + c, T, t)] = make_uint4(a.l[4 * c], a.l[4 * c + 1], a.l[4 * c + 2], a.l[4 * c + 3]);
}
__device__ __forceinline__ Fq ld_pre(const uint4* base, size_t j, size_t T, size_t t) {
    Fq r;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint4 v = base[il<3>(j, c, T, t)];
        r.l[4 * c] = v.x; r.l[4 * c + 1] = v.y; r.l[4 * c + 2] = v.z; r.l[4 * c + 3] = v.w;
    }
    return r;
}
__device__ __forceinline__ void st_pre(uint4* base, size_t j, size_t T, size_t t, const Fq& a) {
#pragma unroll
    for (int c = 0; c < 3; c++) base[il<3>(j, c, T, t)] = make_uint4(a.l[4 * c], a.l[4 * c + 1], a.l[4 * c + 2], a.l[4 * c + 3]);
}

__device__ __forceinline__ FqU fqu_r3() {   // R'^3 mod p: fqu_mul(integer, R'^3) = integer * R'^2
    constexpr u32 m[14] = {0xf63e3ebu, 0xd055de1u, 0x6ff6650u, 0xd6bd950u, 0x9cd510eu, 0x09ed341u, 0x11a3aa6u,
                           0x40b6ca4u, 0x200fa40u, 0x28c4a35u, 0x8a2198cu, 0x956bce5u, 0x96dd52au, 0x5ffu};
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = m[i];
    return r;
}

// a (normalised limbs, value < 64 p) -> the same residue in [0, 3 p), normalised: subtract floor(top limb / ceil(p / 2^364)) p
__device__ __forceinline__ FqU fqu_reduce_small(const FqU& a) {
    const u32 q = a.l[13] / 6884u;                      // p >> 364 = 6883.6: q in {floor(a / p) - 1, floor(a / p)}
    FqU r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        acc += (int64_t)a.l[i] - (int64_t)q * (int64_t)fqu_p(i);
        r.l[i] = (u32)acc & FQU_MASK;
        acc >>= 28;
    }
    r.l[13] = (u32)(acc + (int64_t)a.l[13] - (int64_t)q * (int64_t)fqu_p(13));
    return r;
}
// a - b - c + 8 p, normalised: a, b, c normalised, b + c < 6 p
__device__ __forceinline__ FqU fqu_sub2_norm(const FqU& a, const FqU& b, const FqU& c) {
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = a.l[i] + (fqu_8p_wide(i) - b.l[i] - c.l[i]);
    return fqu_normalize(r);
}

struct WaveDone {   // stop the division steps once every lane of the wave has g == 0
    __device__ bool operator()(bool mine) const { return __all(mine); }
};

// inverse of a U-form value a R' (normalised limbs, < 1.01 p + small): returns a^-1 R' (multiply output)
__device__ __forceinline__ FqU fqu_inv(const FqU& a) {
    Fq w = fqu_pack(a);
    fp_reduce(w);                                             // [0, p)
    Fq i = fq_inv_safegcd_words(w, WaveDone{});               // (a R')^-1 as an integer
    return fqu_mul(fqu_unpack(i), fqu_r3());                  // * R'^2
}

// P1_j <- P1_j + P2_j for the K slots of this thread, `iters` times.  pts1 / pts2: K x 6 chunks x T; pre: K x 3 chunks x T.
template <int K, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_affine_batch(uint4* pts1, const uint4* pts2, uint4* pre, size_t T, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        FqU run = fqu_one();
        for (int j = 0; j < K; j++) {
            FqU x1 = fqu_unpack(ld_fq(pts1, j, 0, T, t)), x2 = fqu_unpack(ld_fq(pts2, j, 0, T, t));
            FqU d = fqu_sub_lazy<4>(x2, x1);                  // x2 - x1 + 4 p  (stored coordinates are < 3 p)
            st_pre(pre, j, T, t, fqu_pack(run));
            run = fqu_mul(run, d);
        }
        FqU inv = fqu_inv(run);
        for (int j = K - 1; j >= 0; j--) {
            FqU x1 = fqu_unpack(ld_fq(pts1, j, 0, T, t)), y1 = fqu_unpack(ld_fq(pts1, j, 3, T, t));
            FqU x2 = fqu_unpack(ld_fq(pts2, j, 0, T, t)), y2 = fqu_unpack(ld_fq(pts2, j, 3, T, t));
            FqU d = fqu_sub_lazy<4>(x2, x1);
            FqU dinv = fqu_mul(inv, fqu_unpack(ld_pre(pre, j, T, t)));
            inv = fqu_mul(inv, d);
            FqU lam = fqu_mul(fqu_sub_lazy<4>(y2, y1), dinv);
            FqU x3 = fqu_reduce_small(fqu_sub2_norm(fqu_sqr(lam), x1, x2));                         // lambda^2 - x1 - x2 (+ 8 p), then < 3 p
            FqU y3 = fqu_reduce_small(fqu_normalize(fqu_sub_lazy<4>(fqu_mul(lam, fqu_sub_lazy<4>(x1, x3)), y1)));
            st_fq(pts1, j, 0, T, t, fqu_pack(x3));
            st_fq(pts1, j, 3, T, t, fqu_pack(y3));
        }
    }
}

// the same additions with the kernel's current XYZZ accumulator (the baseline being challenged): acc_j += P2_j
template <int K, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_xyzz_ref(uint4* pts1, const uint4* pts2, size_t T, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < K; j++) {
        FqU ax = fqu_unpack(ld_fq(pts1, j, 0, T, t)), ay = fqu_unpack(ld_fq(pts1, j, 3, T, t)), azz = fqu_one(), azzz = fqu_one();
        FqU qx = fqu_unpack(ld_fq(pts2, j, 0, T, t)), qy = fqu_unpack(ld_fq(pts2, j, 3, T, t));
        int bad = 0;
        for (int it = 0; it < iters; it++) bad += fqu_xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy) ? 0 : 1;
        st_fq(pts1, j, 0, T, t, fqu_pack(fqu_normalize(fqu_sub_lazy<4>(ax, azz))));
        if (bad == 12345) st_fq(pts1, j, 3, T, t, fqu_pack(ay));
    }
}

// one inversion per thread (cost of the safegcd alone)
__global__ __launch_bounds__(64) void k_inv_only(uint4* pts1, size_t T, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    FqU a = fqu_unpack(ld_fq(pts1, 0, 0, T, t));
    for (int it = 0; it < iters; it++) a = fqu_inv(a);
    st_fq(pts1, 0, 0, T, t, fqu_pack(fqu_normalize(a)));
}

static u64 splitmix(u64& s) {
    s += 0x9E3779B97F4A7C15ull;
    u64 z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int K, int WPE>
static int run_case(size_t T, const std::vector<uint4>& h1, const std::vector<uint4>& h2, uint4* d1, uint4* d2, uint4* dpre, hipEvent_t e0, hipEvent_t e1, const char* dump) {
    const int iters = 8;
    CK(hipMemcpy(d1, h1.data(), (size_t)K * 6 * T * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d2, h2.data(), (size_t)K * 6 * T * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_affine_batch<K, WPE>), dim3((unsigned)(T / 64)), dim3(64), 0, 0, d1, d2, dpre, T, 1);
    CK(hipDeviceSynchronize());
    if (dump) {   // state after ONE addition per slot, for the big-integer check
        std::vector<uint4> out((size_t)K * 6 * T);
        CK(hipMemcpy(out.data(), d1, out.size() * 16, hipMemcpyDeviceToHost));
        FILE* f = fopen(dump, "wb");
        u64 hdr[4] = {(u64)K, (u64)T, 0, 0};
        fwrite(hdr, 8, 4, f);
        fwrite(h1.data(), 16, (size_t)K * 6 * T, f);
        fwrite(h2.data(), 16, (size_t)K * 6 * T, f);
        fwrite(out.data(), 16, out.size(), f);
        fclose(f);
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_affine_batch<K, WPE>), dim3((unsigned)(T / 64)), dim3(64), 0, 0, d1, d2, dpre, T, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double adds = (double)T * K * iters;
    printf("affine batched  K=%-3d wpe=%d  %8.3f ms  %7.3f G additions/s\n", K, WPE, ms, adds / ms / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t T = (size_t)prop.multiProcessorCount * 4 * 64 * 2;   // two waves per SIMD's worth of threads
    const int KMAX = 64;
    printf("device %s CUs %d, %zu threads\n", prop.name, prop.multiProcessorCount, T);
    // random coordinates below 2^376 (< p); the formulas do not need curve points, only x1 != x2
    std::vector<uint4> h1((size_t)KMAX * 6 * T), h2(h1.size());
    u64 s = 0xC0FFEE;
    auto fill = [&](std::vector<uint4>& v) {
        for (size_t e = 0; e < v.size(); e++) {
            u64 a = splitmix(s), b = splitmix(s);
            v[e] = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
        }
        // top word of every coordinate (chunk 2 and 5, component w) < 2^24
        for (size_t j = 0; j < (size_t)KMAX; j++)
            for (int c : {2, 5})
                for (size_t t = 0; t < T; t++) v[((j * 6 + c) * T + t)].w &= 0x00ffffffu;
    };
    fill(h1);
    fill(h2);
    uint4 *d1, *d2, *dpre;
    CK(hipMalloc(&d1, h1.size() * 16));
    CK(hipMalloc(&d2, h2.size() * 16));
    CK(hipMalloc(&dpre, (size_t)KMAX * 3 * T * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* dump = argc > 1 ? argv[1] : nullptr;
    // note: the K-slot layouts of different K overlap in the same buffers (slot-major), which is fine for timing
    if (run_case<8, 2>(T, h1, h2, d1, d2, dpre, e0, e1, nullptr)) return 1;
    if (run_case<16, 2>(T, h1, h2, d1, d2, dpre, e0, e1, dump)) return 1;
    if (run_case<32, 2>(T, h1, h2, d1, d2, dpre, e0, e1, nullptr)) return 1;
    if (run_case<64, 2>(T, h1, h2, d1, d2, dpre, e0, e1, nullptr)) return 1;
    if (run_case<32, 1>(T, h1, h2, d1, d2, dpre, e0, e1, nullptr)) return 1;
    if (run_case<64, 1>(T, h1, h2, d1, d2, dpre, e0, e1, nullptr)) return 1;
    {
        CK(hipMemcpy(d1, h1.data(), (size_t)16 * 6 * T * 16, hipMemcpyHostToDevice));
        const int iters = 16;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_xyzz_ref<16, 2>), dim3((unsigned)(T / 64)), dim3(64), 0, 0, d1, d2, T, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("XYZZ mixed (current kernel's formulas) wpe=2  %8.3f ms  %7.3f G additions/s\n", ms, (double)T * 16 * iters / ms / 1e6);
    }
    {
        const int iters = 16;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_inv_only, dim3((unsigned)(T / 64)), dim3(64), 0, 0, d1, T, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("safegcd inversion alone                      %8.3f ms  %7.3f G inversions/s\n", ms, (double)T * iters / ms / 1e6);
    }
    return 0;
}

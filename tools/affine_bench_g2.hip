// affine_bench_g2.hip -- BATCHED-AFFINE bucket accumulation over Fq2 (G2), prototype + measurement (developer tool, not product).
//
// Question: round 2 built and rejected batched-affine accumulation for G1 only (0.88 x the
// instructions, 0.82 x the speed at K = 64).  In Fq2 the trade is different:
//   XYZZ mixed addition (k_accumulate_u2):  8 M + 2 S over Fq2, fused Y3:            10 332 multiply-adds, 12 951 VALU instructions
//   affine chord with Montgomery's trick:   5 M + 1 S over Fq2 + ONE Fq inversion (safegcd, ~34.5 k instructions) per K additions
// This file measures instructions (under rocprofv3 --pmc SQ_INSTS_VALU) and time per addition of both, in the shape the real
// kernel would have: ONE wave per SIMD, every thread owns K chains (bucket accumulators, affine, in HBM in the kernels' 14 x 28-bit
// limb form -- no packing), the next entry of every chain is a random 256-byte gather from a table, prefix products go through a
// global scratch.  The XYZZ side runs fq2u_xyzz_acc_mixed (the product kernel's formula) on the same gathers with its accumulator
// in registers, as k_accumulate_u2 does.  Go / no-go: affine <= 0.75 x the XYZZ time per addition at K = 64.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../collaborative-zksnark_amd/csrc affine_bench_g2.hip -o affine_bench_g2.bin
// Run:   ./affine_bench_g2.bin [dump.bin]      (tools/affine_check_g2.py dump.bin checks one addition per sampled slot with big integers)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "curve.h"
#include "fqu.h"
#include "lab/fq_safegcd.h"
using namespace czk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// An Fq in limb form occupies 4 x uint4 (14 words + 2 of padding); arrays are interleaved: element (slot j, chunk c, thread t)
__device__ __forceinline__ size_t il(size_t j, int nc, int c, size_t T, size_t t) { return (j * nc + c) * T + t; }

__device__ __forceinline__ FqU ld_u(const uint4* base, size_t j, int nc, int c0, size_t T, size_t t) {
    FqU r;
    uint4 v[4];
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = base[il(j, nc, c0 + c, T, t)];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (4 * c + 0 < 14) r.l[4 * c + 0] = v[c].x;
        if (4 * c + 1 < 14) r.l[4 * c + 1] = v[c].y;
        if (4 * c + 2 < 14) r.l[4 * c + 2] = v[c].z;
        if (4 * c + 3 < 14) r.l[4 * c + 3] = v[c].w;
    }
    return r;
}
__device__ __forceinline__ void st_u(uint4* base, size_t j, int nc, int c0, size_t T, size_t t, const FqU& a) {
#pragma unroll
    for (int c = 0; c < 4; c++)
        base[il(j, nc, c0 + c, T, t)] = make_uint4(a.l[4 * c], a.l[4 * c + 1], 4 * c + 2 < 14 ? a.l[4 * c + 2] : 0u, 4 * c + 3 < 14 ? a.l[4 * c + 3] : 0u);
}
// table entry e: 16 consecutive uint4 = (x.c0, x.c1, y.c0, y.c1), 256 bytes (the "pre-unpacked table" form)
__device__ __forceinline__ FqU ld_tab(const uint4* tab, size_t e, int c0) {
    FqU r;
    const uint4* p = tab + e * 16 + c0;
    uint4 v[4];
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = p[c];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (4 * c + 0 < 14) r.l[4 * c + 0] = v[c].x;
        if (4 * c + 1 < 14) r.l[4 * c + 1] = v[c].y;
        if (4 * c + 2 < 14) r.l[4 * c + 2] = v[c].z;
        if (4 * c + 3 < 14) r.l[4 * c + 3] = v[c].w;
    }
    return r;
}
__device__ __forceinline__ u32 mix(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ FqU fqu_r3() {   // R'^3 mod p: fqu_mul(integer, R'^3) = integer * R'^2
    constexpr u32 m[14] = {0xf63e3ebu, 0xd055de1u, 0x6ff6650u, 0xd6bd950u, 0x9cd510eu, 0x09ed341u, 0x11a3aa6u,
                           0x40b6ca4u, 0x200fa40u, 0x28c4a35u, 0x8a2198cu, 0x956bce5u, 0x96dd52au, 0x5ffu};
    FqU r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.l[i] = m[i];
    return r;
}
// a (normalised limbs, value < 64 p) -> the same residue in [0, 3 p), normalised
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
struct WaveDone {
    __device__ bool operator()(bool mine) const { return __all(mine); }
};
__device__ __noinline__ FqU fqu_inv(const FqU& a) {   // a R' (normalised, < 1.1 p) -> a^-1 R'
    Fq w = fqu_pack(a);
    fp_reduce(w);
    Fq i = fq_inv_safegcd_words(w, WaveDone{});
    return fqu_mul(fqu_unpack(i), fqu_r3());
}
// (a0 + a1 u)^-1 = (a0 - a1 u) / (a0^2 + 5 a1^2); operands multiply outputs (< 1.1 p)
__device__ __forceinline__ Fq2U fq2u_inv(const Fq2U& a) {
    FqU a1x5;
#pragma unroll
    for (int i = 0; i < 14; i++) a1x5.l[i] = 5u * a.c1.l[i];
    FqU nrm = fqu_mul_add(a.c0, a.c0, a.c1, fqu_normalize(a1x5));
    FqU ni = fqu_inv(nrm);
    Fq2U r;
    r.c0 = fqu_mul(a.c0, ni);
    r.c1 = fqu_mul(fqu_sub_lazy<4>(FqU{}, a.c1), ni);
    return r;
}
__device__ __forceinline__ Fq2U fq2u_mulg(const Fq2U& a_lazy, const Fq2U& b_norm) {   // a may be lazy (limbs < 2^30, value < 2^5 p); b a multiply output
    return fq2u_mul_n5(a_lazy, b_norm, fqu_neg5<false>(b_norm.c1));
}
__device__ __forceinline__ Fq2U fq2u_sub4(const Fq2U& a, const Fq2U& b) { return Fq2U{fqu_sub_lazy<4>(a.c0, b.c0), fqu_sub_lazy<4>(a.c1, b.c1)}; }

// acc_j <- acc_j + table[idx(j, t, it)] for the K chains of this thread.  acc: K x 16 chunks x T; pre: K x 8 chunks x T.
template <int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_affine2(uint4* acc, const uint4* tab, u32 tab_mask, uint4* pre, size_t T, int iters) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; it++) {
        Fq2U run{fqu_one(), FqU{}};
        for (int j = 0; j < K; j++) {
            const size_t e = mix((u32)t * 64u + (u32)j + 0x9e3779b9u * (u32)it) & tab_mask;
            Fq2U x1{ld_u(acc, j, 16, 0, T, t), ld_u(acc, j, 16, 4, T, t)};
            Fq2U x2{ld_tab(tab, e, 0), ld_tab(tab, e, 4)};
            Fq2U d = fq2u_sub4(x2, x1);
            st_u(pre, j, 8, 0, T, t, run.c0);
            st_u(pre, j, 8, 4, T, t, run.c1);
            run = fq2u_mulg(d, run);
        }
        Fq2U inv = fq2u_inv(run);
        for (int j = K - 1; j >= 0; j--) {
            const size_t e = mix((u32)t * 64u + (u32)j + 0x9e3779b9u * (u32)it) & tab_mask;
            Fq2U x1{ld_u(acc, j, 16, 0, T, t), ld_u(acc, j, 16, 4, T, t)}, y1{ld_u(acc, j, 16, 8, T, t), ld_u(acc, j, 16, 12, T, t)};
            Fq2U x2{ld_tab(tab, e, 0), ld_tab(tab, e, 4)}, y2{ld_tab(tab, e, 8), ld_tab(tab, e, 12)};
            Fq2U pj{ld_u(pre, j, 8, 0, T, t), ld_u(pre, j, 8, 4, T, t)};
            Fq2U d = fq2u_sub4(x2, x1);
            const FqU n5inv = fqu_neg5<false>(inv.c1);
            Fq2U dinv = fq2u_mul_n5(pj, inv, n5inv);
            inv = fq2u_mul_n5(d, inv, n5inv);
            Fq2U lam = fq2u_mulg(fq2u_sub4(y2, y1), dinv);
            Fq2U l2 = fq2u_sqr(lam);                       // c0 < 18 p, c1 < 2.1 p, normalised
            Fq2U x3;
#pragma unroll
            for (int i = 0; i < 14; i++) {
                x3.c0.l[i] = l2.c0.l[i] + (fqu_8p_wide(i) - x1.c0.l[i] - x2.c0.l[i]);
                x3.c1.l[i] = l2.c1.l[i] + (fqu_8p_wide(i) - x1.c1.l[i] - x2.c1.l[i]);
            }
            x3.c0 = fqu_reduce_small(fqu_normalize(x3.c0));
            x3.c1 = fqu_reduce_small(fqu_normalize(x3.c1));
            Fq2U m = fq2u_mulg(fq2u_sub4(x1, x3), lam);
            Fq2U y3{fqu_reduce_small(fqu_normalize(fqu_sub_lazy<4>(m.c0, y1.c0))), fqu_reduce_small(fqu_normalize(fqu_sub_lazy<4>(m.c1, y1.c1)))};
            st_u(acc, j, 16, 0, T, t, x3.c0);
            st_u(acc, j, 16, 4, T, t, x3.c1);
            st_u(acc, j, 16, 8, T, t, y3.c0);
            st_u(acc, j, 16, 12, T, t, y3.c1);
        }
    }
}

// the product kernel's shape: one chain per thread, XYZZ accumulator in registers, `len` gathered entries added in sequence
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_xyzz2_ref(uint4* acc, const uint4* tab, u32 tab_mask, size_t T, int len) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq2U ax{ld_u(acc, 0, 16, 0, T, t), ld_u(acc, 0, 16, 4, T, t)}, ay{ld_u(acc, 0, 16, 8, T, t), ld_u(acc, 0, 16, 12, T, t)};
    Fq2U azz{fqu_one(), FqU{}}, azzz{fqu_one(), FqU{}};
    int bad = 0;
    for (int it = 0; it < len; it++) {
        const size_t e = mix((u32)t * 64u + 0x9e3779b9u * (u32)it) & tab_mask;
        Fq2U qx{ld_tab(tab, e, 0), ld_tab(tab, e, 4)}, qy{ld_tab(tab, e, 8), ld_tab(tab, e, 12)};
        bad += fq2u_xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy) ? 0 : 1;
    }
    st_u(acc, 0, 16, 0, T, t, ax.c0);
    st_u(acc, 0, 16, 4, T, t, ax.c1);
    st_u(acc, 0, 16, 8, T, t, ay.c0);
    st_u(acc, 0, 16, 12, T, t, azz.c0);
    if (bad == 12345) st_u(acc, 1, 16, 0, T, t, azzz.c1);
}

static u64 splitmix(u64& s) {
    s += 0x9E3779B97F4A7C15ull;
    u64 z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// random field elements in limb form: 13 limbs of 28 bits + a 12-bit top limb (value < 2^376 < p)
static void fill_limbs(std::vector<uint4>& v, u64& s) {
    for (size_t e = 0; e < v.size(); e++) {
        u64 a = splitmix(s), b = splitmix(s);
        v[e] = make_uint4((u32)a & 0x0fffffffu, (u32)(a >> 32) & 0x0fffffffu, (u32)b & 0x0fffffffu, (u32)(b >> 32) & 0x0fffffffu);
    }
}

template <int K>
static int run_case(size_t T, const std::vector<uint4>& hacc, uint4* dacc, const uint4* dtab, u32 mask, uint4* dpre, hipEvent_t e0, hipEvent_t e1, const char* dump,
                    const std::vector<uint4>* htab) {
    const int iters = 8;
    CK(hipMemcpy(dacc, hacc.data(), (size_t)K * 16 * T * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_affine2<K>), dim3((unsigned)(T / 64)), dim3(64), 0, 0, dacc, dtab, mask, dpre, T, 1);
    CK(hipDeviceSynchronize());
    if (dump) {   // state after ONE addition per slot, for the big-integer check
        std::vector<uint4> out((size_t)K * 16 * T);
        CK(hipMemcpy(out.data(), dacc, out.size() * 16, hipMemcpyDeviceToHost));
        FILE* f = fopen(dump, "wb");
        u64 hdr[4] = {(u64)K, (u64)T, (u64)mask, 0};
        fwrite(hdr, 8, 4, f);
        fwrite(hacc.data(), 16, (size_t)K * 16 * T, f);
        fwrite(out.data(), 16, out.size(), f);
        fwrite(htab->data(), 16, htab->size(), f);
        fclose(f);
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_affine2<K>), dim3((unsigned)(T / 64)), dim3(64), 0, 0, dacc, dtab, mask, dpre, T, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double adds = (double)T * K * iters;
    printf("G2 affine batched  K=%-3d  %9.3f ms  %7.4f G additions/s  %7.3f ns per addition and SIMD-lane-slot\n", K, ms, adds / ms / 1e6, ms * 1e6 / (K * iters));
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const size_t T = (size_t)prop.multiProcessorCount * 4 * 64;   // ONE wave per SIMD
    const int KMAX = 64;
    const u32 tab_entries = 1u << 21;                             // 512 MiB of 256-byte entries: gathers miss every cache
    printf("device %s CUs %d, %zu threads (one wave per SIMD), table %u entries\n", prop.name, prop.multiProcessorCount, T, tab_entries);
    std::vector<uint4> hacc((size_t)KMAX * 16 * T), htab((size_t)tab_entries * 16);
    u64 s = 0xC0FFEE;
    fill_limbs(hacc, s);
    fill_limbs(htab, s);
    // top limb (word 13 = chunk 3, component y) 12 bits; padding words (z, w of chunk 3) zero
    for (size_t j = 0; j < (size_t)KMAX; j++)
        for (int f = 0; f < 4; f++)
            for (size_t t = 0; t < T; t++) {
                uint4& v = hacc[(j * 16 + f * 4 + 3) * T + t];
                v.y &= 0xfffu; v.z = 0; v.w = 0;
            }
    for (size_t e = 0; e < tab_entries; e++)
        for (int f = 0; f < 4; f++) {
            uint4& v = htab[e * 16 + f * 4 + 3];
            v.y &= 0xfffu; v.z = 0; v.w = 0;
        }
    uint4 *dacc, *dtab, *dpre;
    CK(hipMalloc(&dacc, hacc.size() * 16));
    CK(hipMalloc(&dtab, htab.size() * 16));
    CK(hipMalloc(&dpre, (size_t)KMAX * 8 * T * 16));
    CK(hipMemcpy(dtab, htab.data(), htab.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* dump = argc > 1 ? argv[1] : nullptr;
    const u32 mask = tab_entries - 1;
    if (run_case<16>(T, hacc, dacc, dtab, mask, dpre, e0, e1, dump, &htab)) return 1;
    if (run_case<32>(T, hacc, dacc, dtab, mask, dpre, e0, e1, nullptr, nullptr)) return 1;
    if (run_case<64>(T, hacc, dacc, dtab, mask, dpre, e0, e1, nullptr, nullptr)) return 1;
    {
        CK(hipMemcpy(dacc, hacc.data(), (size_t)2 * 16 * T * 16, hipMemcpyHostToDevice));
        const int len = 256;
        hipLaunchKernelGGL(k_xyzz2_ref, dim3((unsigned)(T / 64)), dim3(64), 0, 0, dacc, dtab, mask, T, 8);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_xyzz2_ref, dim3((unsigned)(T / 64)), dim3(64), 0, 0, dacc, dtab, mask, T, len);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("G2 XYZZ mixed (k_accumulate_u2's formula, one wave per SIMD)  %9.3f ms  %7.4f G additions/s  %7.3f ns per addition and SIMD-lane-slot\n", ms,
               (double)T * len / ms / 1e6, ms * 1e6 / len);
    }
    return 0;
}

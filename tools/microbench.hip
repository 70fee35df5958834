// microbench.hip -- instruction-rate and field-op throughput probes for gfx950 (developer tool, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../collaborative-zksnark_amd/csrc microbench.hip -o microbench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve.h"
#include "fqu.h"
using namespace czk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k_instr(u32* out, int iters) {
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u64 a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
    u32 x = t * 2654435761u + 1, y = t ^ 0x9e3779b9u;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 8 independent v_mad_u64_u32
#define M(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (MODE == 1) {   // dependent chain
#define M(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
            M(a0) M(a0) M(a0) M(a0) M(a0) M(a0) M(a0) M(a0)
#undef M
        } else if (MODE == 2) {   // v_mul_lo_u32 x8 independent
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(B) : "v"(x));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 3) {   // v_mul_hi_u32
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(B) : "v"(x));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 4) {   // v_add_co_u32 / v_addc pair x4
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B, Cc) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(B), "+v"(Cc) : "v"(x) : "vcc");
            M(b0, b1) M(b2, b3) M(b4, b5) M(b6, b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 5) {   // v_mad_u32_u24 x8
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(B) : "v"(x), "v"(y));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 6) {   // v_fma_f64 x8 independent
            double d0 = __longlong_as_double(a0), d1 = __longlong_as_double(a1), d2 = __longlong_as_double(a2), d3 = __longlong_as_double(a3);
            double d4 = __longlong_as_double(a4), d5 = __longlong_as_double(a5), d6 = __longlong_as_double(a6), d7 = __longlong_as_double(a7);
            double xx = __longlong_as_double(a0 | 0x3ff0000000000000ull);
#define M(D) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(D) : "v"(xx));
            M(d0) M(d1) M(d2) M(d3) M(d4) M(d5) M(d6) M(d7)
#undef M
            a0 = __double_as_longlong(d0); a1 = __double_as_longlong(d1); a2 = __double_as_longlong(d2); a3 = __double_as_longlong(d3);
            a4 = __double_as_longlong(d4); a5 = __double_as_longlong(d5); a6 = __double_as_longlong(d6); a7 = __double_as_longlong(d7);
        } else if (MODE == 7) {   // mad + addc pair (the Comba step) x8 on 2 accumulators
            u32 h0 = (u32)a7, h1 = (u32)a6;
#define M(A, H) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(A), "+v"(H) : "v"(x), "v"(y) : "vcc");
            M(a0, h0) M(a1, h1) M(a0, h0) M(a1, h1) M(a0, h0) M(a1, h1) M(a0, h0) M(a1, h1)
#undef M
            a7 = h0; a6 = h1;
        }
    }
    out[t] = (u32)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

template <class F>
__global__ void k_mul(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    F a = FieldIO<F>::load(data + FieldIO<F>::W64 * t), b = a;
    for (int i = 0; i < iters; i++) { a = f_mul(a, b); b = f_mul(b, a); }
    FieldIO<F>::store(data + FieldIO<F>::W64 * t, f_add(a, b));
}
__global__ void k_mul_fr(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Fr a = fp_load<FrParams>(data + 4 * t), b = a;
    for (int i = 0; i < iters; i++) { a = fp_mul(a, b); b = fp_mul(b, a); }
    fp_store<FrParams>(data + 4 * t, fp_add(a, b));
}
template <int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_madd(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Affine<Fq> p = aff_load<Fq>(data + 12 * t);
    Jac<Fq> acc{p.x, p.y, Fq::one()};
    acc = jac_double(acc);
    for (int i = 0; i < iters; i++) acc = jac_add_mixed(acc, p, false);
    jac_store<Fq>(data + 18 * t, acc);
}
__global__ void k_mul_u(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    FqU a = fqu_unpack(fp_load<FqParams>(data + 6 * t)), b = a;
    for (int i = 0; i < iters; i++) { a = fqu_mul(a, b); b = fqu_mul(b, a); }
    fp_store<FqParams>(data + 6 * t, fqu_pack(fqu_normalize(fqu_sub_lazy<4>(a, b))));
}
template <int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_madd_u(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    FqU qx = fqu_unpack(fp_load<FqParams>(data + 12 * t)), qy = fqu_unpack(fp_load<FqParams>(data + 12 * t + 6));
    FqU ax = qy, ay = qx, azz = fqu_one(), azzz = fqu_one();
    int bad = 0;
    for (int i = 0; i < iters; i++) bad += fqu_xyzz_acc_mixed(ax, ay, azz, azzz, qx, qy) ? 0 : 1;
    fp_store<FqParams>(data + 18 * t, fqu_pack(fqu_normalize(fqu_sub_lazy<4>(ax, azz))));
    if (bad == 12345) data[0] = 1;
}
template <int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_madd_x(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Affine<Fq> p = aff_load<Fq>(data + 12 * t);
    Fq ax = p.y, ay = p.x, azz = Fq::one(), azzz = Fq::one();
    for (int i = 0; i < iters; i++) xyzz_acc_mixed(ax, ay, azz, azzz, p.x, p.y);
    fp_store<FqParams>(data + 18 * t, fp_add(ax, azz));
}
template <int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_madd2(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Affine<Fq2> p = aff_load<Fq2>(data + 24 * t);
    Jac<Fq2> acc{p.x, p.y, Fq2::one()};
    acc = jac_double(acc);
    for (int i = 0; i < iters; i++) acc = jac_add_mixed(acc, p, false);
    jac_store<Fq2>(data + 36 * t, acc);
}
template <int WPE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_mul2(u64* data, int iters) {
    size_t t = threadIdx.x + (size_t)blockIdx.x * blockDim.x;
    Fq2 a = FieldIO<Fq2>::load(data + 12 * t), b = a;
    for (int i = 0; i < iters; i++) { a = f_mul(a, b); b = f_mul(b, a); }
    FieldIO<Fq2>::store(data + 12 * t, f_add(a, b));
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    u32* out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"v_mad_u64_u32 indep", "v_mad_u64_u32 dep", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co+v_addc (2 instr)", "v_mad_u32_u24", "v_fma_f64", "mad_u64+addc (2 instr)"};
    const int iters = 4000;
#define RUN(MODE) { hipLaunchKernelGGL(k_instr<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); \
    hipLaunchKernelGGL(k_instr<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    double ops = (double)blocks * threads * iters * 8; printf("%-28s %8.3f ms  %8.2f Gops/s(lane)  => %.2f cycles/wave-instr/SIMD @2.4GHz\n", names[MODE], ms, ops / ms / 1e6, (double)prop.multiProcessorCount * 4 * 2.4e9 / (ops / 64 / (ms / 1e3))); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)

    // field multiply throughput
    size_t n = (size_t)blocks * threads;
    u64* data; CK(hipMalloc(&data, n * 36 * 8));
    std::vector<u64> h(n * 36);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i * 0x9e3779b97f4a7c15ull) >> 9;
    CK(hipMemcpy(data, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    auto timeit = [&](const char* name, auto launch, double ops_per_thread, size_t nthreads) {
        launch(4); hipDeviceSynchronize(); hipEventRecord(e0); launch(100); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms  %8.2f G field-mul-equiv/s\n", name, ms, ops_per_thread * 100 * nthreads / ms / 1e6);
    };
    timeit("Fr mul (256thr x8/CU)", [&](int it) { hipLaunchKernelGGL(k_mul_fr, dim3(blocks), dim3(threads), 0, 0, data, it); }, 2, n);
    timeit("Fq mul (256thr x8/CU)", [&](int it) { hipLaunchKernelGGL(k_mul<Fq>, dim3(blocks), dim3(threads), 0, 0, data, it); }, 2, n);
    timeit("FqU mul (14x28 unsat)", [&](int it) { hipLaunchKernelGGL(k_mul_u, dim3(blocks), dim3(threads), 0, 0, data, it); }, 2, n);
    timeit("G1 xyzz madd sat wpe2 (10M)", [&](int it) { hipLaunchKernelGGL(k_madd_x<2>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 10, n);
    timeit("G1 xyzz madd UNSAT wpe2", [&](int it) { hipLaunchKernelGGL(k_madd_u<2>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 10, n);
    timeit("G1 xyzz madd UNSAT wpe3", [&](int it) { hipLaunchKernelGGL(k_madd_u<3>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 10, n);
    timeit("Fq2 mul (=3 Fq mul)", [&](int it) { hipLaunchKernelGGL(k_mul<Fq2>, dim3(blocks), dim3(threads), 0, 0, data, it); }, 6, n);
    timeit("Fq2 mul wpe2", [&](int it) { hipLaunchKernelGGL(k_mul2<2>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 6, n);
    timeit("Fq2 mul wpe3", [&](int it) { hipLaunchKernelGGL(k_mul2<3>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 6, n);
    timeit("G2 madd wpe1", [&](int it) { hipLaunchKernelGGL(k_madd2<1>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 33, n);
    timeit("G2 madd wpe2", [&](int it) { hipLaunchKernelGGL(k_madd2<2>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 33, n);
    timeit("G1 madd wpe1", [&](int it) { hipLaunchKernelGGL(k_madd<1>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 11, n);
    timeit("G1 madd wpe2", [&](int it) { hipLaunchKernelGGL(k_madd<2>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 11, n);
    timeit("G1 madd wpe3", [&](int it) { hipLaunchKernelGGL(k_madd<3>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 11, n);
    timeit("G1 madd wpe4", [&](int it) { hipLaunchKernelGGL(k_madd<4>, dim3(blocks * 2), dim3(128), 0, 0, data, it); }, 11, n);
    return 0;
}

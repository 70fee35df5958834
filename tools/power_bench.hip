// power_bench.hip -- board power under sustained loops of the candidate multiply instructions (developer tool; run by tools/power_bench.sh,
// which samples `rocm-smi --showpower` beside it).  The bucket kernels are bound by the power management, so the figure of merit of a limb
// multiplier is energy per bit^2 of partial product, not issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 power_bench.hip -o power_bench.bin ;  usage: power_bench.bin MODE SECONDS
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <chrono>
typedef uint32_t u32;
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// 64 instructions per asm statement, eight independent destinations, operands that look like limbs
#define R8(S) S(16, 17) S(20, 21) S(24, 25) S(28, 29) S(32, 33) S(36, 37) S(40, 41) S(44, 45)
#define CLOB "vcc", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45", "v50", "v51", "v52", "v53"
template <int MODE>
__global__ void k_power(u32* out, int iters) {
    u32 t = threadIdx.x + blockIdx.x * blockDim.x, s = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // v_mad_u64_u32, 28-bit operands
#define S(A, B) "v_mad_u64_u32 v[" #A ":" #B "], vcc, v50, v51, v[" #A ":" #B "]\n v_mad_u64_u32 v[" #A ":" #B "], vcc, v52, v53, v[" #A ":" #B "]\n"
            asm volatile("v_and_b32 v50, 0xfffffff, %0\n v_and_b32 v51, 0xfffffff, %1\n v_xor_b32 v52, 0x5a5a5a5, v50\n v_xor_b32 v53, 0xa5a5a5a, v51\n" R8(S) R8(S) R8(S) R8(S) :: "v"(t * 2654435761u + i), "v"(t * 40503u + 77u * i) : CLOB);
#undef S
        }
        if (MODE == 1) {   // v_mul_u32_u24 + v_mul_hi_u32_u24 (a 24 x 24 -> 48-bit product in two full-rate instructions)
#define S(A, B) "v_mul_u32_u24 v" #A ", v50, v51\n v_mul_hi_u32_u24 v" #B ", v50, v51\n v_mul_u32_u24 v" #A ", v52, v53\n v_mul_hi_u32_u24 v" #B ", v52, v53\n"
            asm volatile("v_and_b32 v50, 0xffffff, %0\n v_and_b32 v51, 0xffffff, %1\n v_xor_b32 v52, 0x5a5a5a, v50\n v_xor_b32 v53, 0xa5a5a5, v51\n" R8(S) R8(S) R8(S) R8(S) :: "v"(t * 2654435761u + i), "v"(t * 40503u + 77u * i) : CLOB);
#undef S
        }
        if (MODE == 2) {   // v_mad_u32_u24 (24 x 24 low half + 32-bit addend)
#define S(A, B) "v_mad_u32_u24 v" #A ", v50, v51, v" #A "\n v_mad_u32_u24 v" #B ", v52, v53, v" #B "\n"
            asm volatile("v_and_b32 v50, 0xffffff, %0\n v_and_b32 v51, 0xffffff, %1\n v_xor_b32 v52, 0x5a5a5a, v50\n v_xor_b32 v53, 0xa5a5a5, v51\n" R8(S) R8(S) R8(S) R8(S) :: "v"(t * 2654435761u + i), "v"(t * 40503u + 77u * i) : CLOB);
#undef S
        }
        if (MODE == 3) {   // v_fma_f64 (53-bit mantissa: exact sums of 26 x 26-bit products)
#define S(A, B) "v_fma_f64 v[" #A ":" #B "], v[50:51], v[52:53], v[" #A ":" #B "]\n v_fma_f64 v[" #A ":" #B "], v[52:53], v[50:51], v[" #A ":" #B "]\n"
            asm volatile("v_cvt_f64_u32 v[50:51], %0\n v_cvt_f64_u32 v[52:53], %1\n" R8(S) R8(S) R8(S) R8(S) :: "v"((t * 2654435761u + i) & 0x3ffffffu), "v"((t * 40503u + 77u * i) & 0x3ffffffu) : CLOB);
#undef S
        }
        if (MODE == 4) {   // v_mul_lo_u32 + v_mul_hi_u32 (32 x 32 -> 64 in two quarter-rate instructions)
#define S(A, B) "v_mul_lo_u32 v" #A ", v50, v51\n v_mul_hi_u32 v" #B ", v50, v51\n v_mul_lo_u32 v" #A ", v52, v53\n v_mul_hi_u32 v" #B ", v52, v53\n"
            asm volatile("v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n v_xor_b32 v52, 0x5a5a5a5a, v50\n v_xor_b32 v53, 0xa5a5a5a5, v51\n" R8(S) R8(S) R8(S) R8(S) :: "v"(t * 2654435761u + i), "v"(t * 40503u + 77u * i) : CLOB);
#undef S
        }
        if (MODE == 5) {   // v_dot4_i32_i8 (four 8 x 8 products + 32-bit addend, full rate)
#define S(A, B) "v_dot4_i32_i8 v" #A ", v50, v51, v" #A "\n v_dot4_i32_i8 v" #B ", v52, v53, v" #B "\n"
            asm volatile("v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n v_xor_b32 v52, 0x5a5a5a5a, v50\n v_xor_b32 v53, 0xa5a5a5a5, v51\n" R8(S) R8(S) R8(S) R8(S) :: "v"(t * 2654435761u + i), "v"(t * 40503u + 77u * i) : CLOB);
#undef S
        }
    }
    asm volatile("v_add_u32 %0, v16, v20\n v_add_u32 %0, %0, v44" : "=v"(s) :: "v16", "v20", "v44");
    out[t] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 2000;
    u32* out;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    const char* names[] = {"v_mad_u64_u32 (28-bit limbs)", "v_mul_u32_u24 + v_mul_hi_u32_u24", "v_mad_u32_u24", "v_fma_f64", "v_mul_lo_u32 + v_mul_hi_u32", "v_dot4_i32_i8"};
    const double bit2[] = {784, 576.0 / 2, 576.0 / 2, 676, 1024.0 / 2, 256};   // partial-product bits^2 per INSTRUCTION (low-half-only forms count half)
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int j = 0; j < 8; j++) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_power<0>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
                case 1: hipLaunchKernelGGL(k_power<1>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
                case 2: hipLaunchKernelGGL(k_power<2>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
                case 3: hipLaunchKernelGGL(k_power<3>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
                case 4: hipLaunchKernelGGL(k_power<4>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
                default: hipLaunchKernelGGL(k_power<5>, dim3(blocks), dim3(threads), 0, 0, out, iters); break;
            }
        }
        launches += 8;
        CK(hipDeviceSynchronize());
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double instr = (double)launches * blocks * threads * iters * 64;
    printf("mode %d %-36s %6.2f s  %7.2f T instr/s  %8.2f P bit^2/s\n", mode, names[mode], el, instr / el / 1e12, instr / el * bit2[mode] / 1e15);
    return 0;
}

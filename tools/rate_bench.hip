// rate_bench.hip -- issue rates of the integer instructions the field arithmetic is built from (developer tool).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 rate_bench.hip -o rate_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k_rate(u32* out, int iters) {
    u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u64 a0 = t * 0x9e3779b97f4a7c15ull + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    u32 x = t * 2654435761u + 1, y = t ^ 0x9e3779b9u;
    for (int i = 0; i < iters; i++) {
#define R8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
        if (MODE == 0) {
#define M(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
            R8(M)
#undef M
        } else if (MODE == 1) {
#define M(A) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(A));
            R8(M)
#undef M
        } else if (MODE == 2) {
#define M(A) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(A) : "v"(a7 | 1));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a0)
#undef M
        } else if (MODE == 3) {   // the 32-bit pair that replaces one 64-bit right shift
            u32 l0 = (u32)a0, h0 = (u32)(a0 >> 32), l1 = (u32)a1, h1 = (u32)(a1 >> 32), l2 = (u32)a2, h2 = (u32)(a2 >> 32), l3 = (u32)a3, h3 = (u32)(a3 >> 32);
#define M(L, H) asm volatile("v_alignbit_b32 %0, %1, %0, 29\n v_lshrrev_b32 %1, 29, %1" : "+v"(L), "+v"(H));
            M(l0, h0) M(l1, h1) M(l2, h2) M(l3, h3) M(l0, h0) M(l1, h1) M(l2, h2) M(l3, h3)
#undef M
            a0 = l0 | ((u64)h0 << 32); a1 = l1 | ((u64)h1 << 32); a2 = l2 | ((u64)h2 << 32); a3 = l3 | ((u64)h3 << 32);
        } else if (MODE == 4) {
#define M(A) asm volatile("v_ashrrev_i64 %0, 30, %0" : "+v"(A));
            R8(M)
#undef M
        } else if (MODE == 5) {
#define M(A) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
            R8(M)
#undef M
        } else if (MODE == 6) {   // 32-bit and / sub (simple VALU)
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B) asm volatile("v_and_b32 %0, %1, %0" : "+v"(B) : "v"(x));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 7) {   // v_add3_u32
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
#define M(B) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(B) : "v"(x), "v"(y));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 8) {   // v_mad_u64_u32 interleaved with s_nop (what one-asm-statement-per-product costs)
#define M(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0" : "+v"(A) : "v"(x), "v"(y) : "vcc");
            R8(M)
#undef M
        } else if (MODE == 9) {   // v_mul_u32_u24 + v_mul_hi_u32_u24 pair
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3;
            u32 c0 = (u32)a4, c1 = (u32)a5, c2 = (u32)a6, c3 = (u32)a7;
#define M(B, Cc) asm volatile("v_mul_u32_u24 %0, %0, %2\n v_mul_hi_u32_u24 %1, %1, %2" : "+v"(B), "+v"(Cc) : "v"(x));
            M(b0, c0) M(b1, c1) M(b2, c2) M(b3, c3)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = c0; a5 = c1; a6 = c2; a7 = c3;
        } else if (MODE == 10) {   // v_sad_u32 d = |s - b| + a with the constant in an SGPR: the one-instruction form of a + (C - b)
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
            const u32 cst = 0x40000000u + (u32)iters;
#define M(B) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(B) : "s"(cst), "v"(x));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (MODE == 15) {   // eight chains with eight different, random-looking operand pairs: the multiplier array toggles on every issue (data-dependent power)
            u32 x1 = x * 0x85ebca6bu + 0x1234567u, y1 = y * 0xc2b2ae35u + 0x89abcdeu, x2 = x1 * 0x27d4eb2fu + 7u, y2 = y1 * 0x165667b1u + 3u, x3 = ~x, y3 = ~y1;
            x1 &= 0x0fffffffu; y1 &= 0x0fffffffu; x2 &= 0x0fffffffu; y2 &= 0x0fffffffu; x3 &= 0x0fffffffu; y3 &= 0x0fffffffu;
#define M(A, X, Y) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y) : "vcc");
            M(a0, x1, y1) M(a1, x2, y2) M(a2, x3, y3) M(a3, x1, y2) M(a4, x2, y3) M(a5, x3, y1) M(a6, x1, y3) M(a7, x2, y1)
#undef M
        } else if (MODE == 11) {   // the two-instruction form it would replace
            u32 b0 = (u32)a0, b1 = (u32)a1, b2 = (u32)a2, b3 = (u32)a3, b4 = (u32)a4, b5 = (u32)a5, b6 = (u32)a6, b7 = (u32)a7;
            const u32 cst = 0x40000000u + (u32)iters;
#define M(B) asm volatile("v_sub_u32 %1, %2, %1\n v_add_u32 %0, %0, %1" : "+v"(B), "+v"(y) : "s"(cst));
            M(b0) M(b1) M(b2) M(b3) M(b4) M(b5) M(b6) M(b7)
#undef M
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        }
    }
    out[t] = (u32)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + (u32)((a0 ^ a5) >> 32);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 4000;
    u32* out;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[] = {"v_mad_u64_u32", "v_lshrrev_b64", "v_lshl_add_u64", "v_alignbit_b32 + v_lshrrev_b32 (2 instr)", "v_ashrrev_i64", "v_mad_i64_i32", "v_and_b32", "v_add3_u32",
                           "v_mad_u64_u32 + s_nop 0 (2 slots)", "v_mul_u32_u24 + v_mul_hi_u32_u24 (2 instr)", "v_sad_u32 (sgpr, v, v)", "v_sub_u32 + v_add_u32 (2 instr)"};
#define RUN(MODE) { hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); \
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
    double ops = (double)blocks * threads * iters * 8; printf("%-44s %8.3f ms  %9.1f G lane-ops/s (8 per loop iteration)\n", names[MODE], ms, ops / ms / 1e6); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    // Sustained v_mad_u64_u32 rate: the figures above come from 0.5 ms bursts, which finish before the power management reacts.  The
    // bucket kernels run for tens of milliseconds back to back, so this is the rate they can be priced against: 40 windows of 16
    // launches (~10 ms each), per-window rate.
    {
        const int windows = 40, per = 16;
        hipEvent_t ev[windows + 1];
        for (int i = 0; i <= windows; i++) CK(hipEventCreate(&ev[i]));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ev[0]));
        for (int w = 0; w < windows; w++) {
            for (int j = 0; j < per; j++) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(threads), 0, 0, out, iters);
            CK(hipEventRecord(ev[w + 1]));
        }
        CK(hipEventSynchronize(ev[windows]));
        printf("sustained v_mad_u64_u32 (8 independent chains per lane, %d waves per SIMD), T lane-ops/s per ~10 ms window:\n ", blocks * threads / 64 / (prop.multiProcessorCount * 4));
        double last = 0;
        for (int w = 0; w < windows; w++) {
            float ms;
            CK(hipEventElapsedTime(&ms, ev[w], ev[w + 1]));
            last = (double)blocks * threads * iters * 8 * per / ms / 1e9;
            printf(" %.1f", last);
        }
        printf("\nsustained_mad_u64_u32_T_per_s %.2f\n", last);
    }
    // The same sustained run with operands that look like field limbs (28 random bits, a different pair per chain) instead of one pair
    // of registers for every multiply-add: switching activity in the multiplier is what the power management sees
    {
        const int windows = 24, per = 16;
        hipEvent_t ev[windows + 1];
        for (int i = 0; i <= windows; i++) CK(hipEventCreate(&ev[i]));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ev[0]));
        for (int w = 0; w < windows; w++) {
            for (int j = 0; j < per; j++) hipLaunchKernelGGL(k_rate<15>, dim3(blocks), dim3(threads), 0, 0, out, iters);
            CK(hipEventRecord(ev[w + 1]));
        }
        CK(hipEventSynchronize(ev[windows]));
        printf("sustained v_mad_u64_u32, random 28-bit operands, a different pair per chain, T lane-ops/s per ~10 ms window:\n ");
        for (int w = 0; w < windows; w++) {
            float ms;
            CK(hipEventElapsedTime(&ms, ev[w], ev[w + 1]));
            printf(" %.1f", (double)blocks * threads * iters * 8 * per / ms / 1e9);
        }
        printf("\n");
    }
    // (Latency / occupancy questions are answered by tools/bank_bench.hip, whose blocks are single asm statements: between SEPARATE asm
    // statements the compiler inserts an s_nop, which makes a one-accumulator loop of this file look latency-bound when it is not.)
    return 0;
}

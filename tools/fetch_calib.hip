// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access pattern of the MSM accumulate kernels.
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern".  k_accumulate_te / k_accumulate_u2 gather 192-byte table entries (three aligned
// 64-byte sectors, 16-byte loads) at effectively random indices of a multi-GB table; this program does exactly that with a KNOWN count:
//   k_gather : every thread reads `per_thread` entries of 192 B at hashed indices spread over a `table_gb` GB table (12 x 16-byte loads each)
//   k_stream : every thread reads 16 B per iteration, coalesced, once over the whole table (the guide's pattern: expected factor 0.5)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; tools/fetch_calib_summary.py divides the counter by the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("FAILED %s: %s\n", #x, hipGetErrorString(e));                   \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_gather(const ulonglong2* table, size_t entries, unsigned per_thread, unsigned long long* sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (unsigned k = 0; k < per_thread; k++) {
        const size_t e = mix(tid * 0x9E3779B97F4A7C15ull + k + 1) % entries;
        const ulonglong2* p = table + e * 12;          // 192 B = 12 x 16 B, three 64-byte sectors
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const ulonglong2 v = p[j];
            acc ^= v.x + v.y;
        }
    }
    if (acc == 0x123456789abcdefull) sink[tid & 1023] = acc;   // keeps the loads alive, (almost) never writes
}

__global__ void k_stream(const ulonglong2* table, size_t n16, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = table[i];
        acc ^= v.x + v.y;
    }
    if (acc == 0x123456789abcdefull) sink[threadIdx.x] = acc;
}

__global__ void k_fill(unsigned long long* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = mix(i + 1);
}

int main(int argc, char** argv) {
    const double table_gb = argc > 1 ? atof(argv[1]) : 4.0;
    const unsigned per_thread = argc > 2 ? (unsigned)atoi(argv[2]) : 64;
    const size_t entries = (size_t)(table_gb * 1e9 / 192);
    const size_t bytes = entries * 192;
    void *table = nullptr, *sink = nullptr;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&sink, 8192));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned long long*)table, bytes / 8);
    CK(hipDeviceSynchronize());
    const unsigned blocks = 8192, threads = 256;       // 2 M threads: 32 waves per SIMD's worth of work, like a bucket kernel
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, (const ulonglong2*)table, entries, per_thread, (unsigned long long*)sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double gb = (double)blocks * threads * per_thread * 192 / 1e9;
        printf("{\"kernel\": \"k_gather\", \"entries_read\": %zu, \"bytes\": %.0f, \"table_bytes\": %zu, \"ms\": %.3f, \"gb_per_s\": %.1f}\n",
               (size_t)blocks * threads * per_thread, gb * 1e9, bytes, ms, gb / (ms / 1e3));
    }
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const ulonglong2*)table, bytes / 16, (unsigned long long*)sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"k_stream\", \"bytes\": %zu, \"ms\": %.3f, \"gb_per_s\": %.1f}\n", bytes, ms, bytes / 1e9 / (ms / 1e3));
    }
    return 0;
}

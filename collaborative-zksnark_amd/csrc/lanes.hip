// lanes.hip -- device-resident share lanes behind opaque handles (czk.h "device-resident share lanes").
//
// The reference keeps its share vectors in `Vec`s that live across the whole witness map and feed the MSM directly
// (mpc-snarks/src/groth/r1cs_to_qap.rs:85-110, mpc-snarks/src/groth/prover.rs:104).  A caller that owns no HIP allocator
// (the Rust shim, include/czk.hpp) gets the same lifetime on the GPU with a czk_lanes handle: allocate once, upload the witness
// lanes once, run every transform / pointwise step / MSM on czk_lanes_data() addresses with CZK_MEM_DEVICE, download only what
// the host needs.  Host code only (no kernels): transfers are staged through two pinned chunks owned by the context.
#include "czk_internal.h"

#include <string.h>

struct czk_lanes {
    int device = 0;
    size_t lanes = 0, len = 0;
    uint64_t* p = nullptr;
};

namespace czk {

constexpr size_t XFER_CHUNK = (size_t)16 << 20;   // bytes per pinned staging chunk

static int xfer_init(czk_ctx* ctx) {
    if (ctx->xfer_pinned[0]) return CZK_OK;
    for (int k = 0; k < 2; k++) {
        CZK_HIP(ctx, hipHostMalloc((void**)&ctx->xfer_pinned[k], XFER_CHUNK, hipHostMallocDefault));
        CZK_HIP(ctx, hipEventCreateWithFlags(&ctx->xfer_ev[k], hipEventDisableTiming));
        ctx->xfer_busy[k] = false;
    }
    return CZK_OK;
}

void xfer_destroy(czk_ctx* ctx) {
    for (int k = 0; k < 2; k++) {
        if (ctx->xfer_pinned[k]) (void)hipHostFree(ctx->xfer_pinned[k]);
        if (ctx->xfer_ev[k]) (void)hipEventDestroy(ctx->xfer_ev[k]);
        ctx->xfer_pinned[k] = nullptr;
        ctx->xfer_ev[k] = nullptr;
    }
}

// the chunk's last DMA (either direction) has finished
static int xfer_wait(czk_ctx* ctx, int k) {
    if (ctx->xfer_busy[k]) {
        CZK_HIP(ctx, hipEventSynchronize(ctx->xfer_ev[k]));
        ctx->xfer_busy[k] = false;
    }
    return CZK_OK;
}

// pageable host -> device in stream order; returns when `host` has been read completely
int upload_pageable(czk_ctx* ctx, void* dev, const void* host, size_t bytes) {
    CZK_TRY(xfer_init(ctx));
    int k = 0;
    for (size_t off = 0; off < bytes; off += XFER_CHUNK, k ^= 1) {
        const size_t nb = bytes - off < XFER_CHUNK ? bytes - off : XFER_CHUNK;
        CZK_TRY(xfer_wait(ctx, k));
        memcpy(ctx->xfer_pinned[k], (const char*)host + off, nb);
        CZK_HIP(ctx, hipMemcpyAsync((char*)dev + off, ctx->xfer_pinned[k], nb, hipMemcpyHostToDevice, ctx->stream));
        CZK_HIP(ctx, hipEventRecord(ctx->xfer_ev[k], ctx->stream));
        ctx->xfer_busy[k] = true;
    }
    return CZK_OK;
}

// device -> pageable host, blocking; the copy out of chunk k overlaps the DMA into chunk k ^ 1
int download_pageable(czk_ctx* ctx, void* host, const void* dev, size_t bytes) {
    CZK_TRY(xfer_init(ctx));
    int k = 0;
    size_t prev_off = 0, prev_nb = 0;
    for (size_t off = 0; off < bytes; off += XFER_CHUNK, k ^= 1) {
        const size_t nb = bytes - off < XFER_CHUNK ? bytes - off : XFER_CHUNK;
        CZK_TRY(xfer_wait(ctx, k));
        CZK_HIP(ctx, hipMemcpyAsync(ctx->xfer_pinned[k], (const char*)dev + off, nb, hipMemcpyDeviceToHost, ctx->stream));
        CZK_HIP(ctx, hipEventRecord(ctx->xfer_ev[k], ctx->stream));
        ctx->xfer_busy[k] = true;
        if (prev_nb) {
            CZK_TRY(xfer_wait(ctx, k ^ 1));
            memcpy((char*)host + prev_off, ctx->xfer_pinned[k ^ 1], prev_nb);
        }
        prev_off = off;
        prev_nb = nb;
    }
    if (prev_nb) {
        CZK_TRY(xfer_wait(ctx, k ^ 1));
        memcpy((char*)host + prev_off, ctx->xfer_pinned[k ^ 1], prev_nb);
    }
    return CZK_OK;
}

// ranges are contiguous runs of the lane-major array: one that starts in lane `lane` may run on into the following lanes
static bool in_range(const czk_lanes* l, size_t lane, size_t elem, size_t n) {
    return l && lane < l->lanes && elem <= l->len && n <= (l->lanes - lane) * l->len - elem;
}

}  // namespace czk

using namespace czk;

extern "C" int czk_lanes_alloc(czk_ctx* ctx, size_t lanes, size_t len, czk_lanes** out) {
    if (!ctx || !out) return ctx ? set_err(ctx, CZK_ERR_ARG, "null lanes_alloc argument") : CZK_ERR_ARG;
    *out = nullptr;
    if (len && lanes > ((size_t)1 << 58) / len) return set_err(ctx, CZK_ERR_SIZE, "lanes x len overflows");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    czk_lanes* l = new czk_lanes();
    l->device = ctx->device;
    l->lanes = lanes;
    l->len = len;
    const size_t bytes = lanes * len * 32;
    if (bytes) {
        hipError_t e = hipMalloc((void**)&l->p, bytes);
        if (e != hipSuccess) {
            delete l;
            return set_err(ctx, CZK_ERR_NOMEM, std::string("hipMalloc share lanes: ") + hipGetErrorString(e));
        }
        e = hipMemsetAsync(l->p, 0, bytes, ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(l->p);
            delete l;
            return set_err(ctx, CZK_ERR_HIP, std::string("hipMemsetAsync share lanes: ") + hipGetErrorString(e));
        }
    }
    *out = l;
    return CZK_OK;
}

extern "C" void czk_lanes_free(czk_lanes* l) {
    if (!l) return;
    (void)hipSetDevice(l->device);
    if (l->p) (void)hipFree(l->p);   // hipFree waits for the device: no kernel still reads the lanes
    delete l;
}

extern "C" size_t czk_lanes_count(const czk_lanes* l) { return l ? l->lanes : 0; }
extern "C" size_t czk_lanes_len(const czk_lanes* l) { return l ? l->len : 0; }
extern "C" uint64_t* czk_lanes_data(const czk_lanes* l, size_t lane, size_t elem) {
    if (!l || !l->p || lane >= l->lanes || elem >= l->len) return nullptr;
    return l->p + 4 * (lane * l->len + elem);
}

extern "C" int czk_lanes_upload(czk_ctx* ctx, czk_lanes* dst, size_t lane, size_t elem, const uint64_t* host, size_t n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!in_range(dst, lane, elem, n) || (n && !host)) return set_err(ctx, CZK_ERR_ARG, "czk_lanes_upload: range outside the lanes / null host pointer");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    return upload_pageable(ctx, dst->p + 4 * (lane * dst->len + elem), host, n * 32);
}

extern "C" int czk_lanes_download(czk_ctx* ctx, const czk_lanes* src, size_t lane, size_t elem, uint64_t* host, size_t n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!in_range(src, lane, elem, n) || (n && !host)) return set_err(ctx, CZK_ERR_ARG, "czk_lanes_download: range outside the lanes / null host pointer");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    return download_pageable(ctx, host, src->p + 4 * (lane * src->len + elem), n * 32);
}

extern "C" int czk_lanes_download_deferred(czk_ctx* ctx, const czk_lanes* src, size_t lane, size_t elem, uint64_t* host, size_t n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!in_range(src, lane, elem, n) || (n && !host)) return set_err(ctx, CZK_ERR_ARG, "czk_lanes_download_deferred: range outside the lanes / null host pointer");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    // into the pinned result ring on the context's stream; delivered to `host` with the MSM results (czk_ctx_wait_mark / czk_ctx_sync)
    char* pinned = nullptr;
    CZK_TRY(msm_pinned_take(ctx, n * 32, &pinned));
    CZK_HIP(ctx, hipMemcpyAsync(pinned, src->p + 4 * (lane * src->len + elem), n * 32, hipMemcpyDeviceToHost, ctx->stream));
    MsmPending pend{pinned, host, n * 32};
    pend.on_stream = true;
    ctx->msm_pending.push_back(pend);
    return CZK_OK;
}

extern "C" int czk_lanes_copy(czk_ctx* ctx, czk_lanes* dst, size_t dst_lane, size_t dst_elem, const czk_lanes* src, size_t src_lane, size_t src_elem,
                              size_t n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!in_range(dst, dst_lane, dst_elem, n) || !in_range(src, src_lane, src_elem, n)) return set_err(ctx, CZK_ERR_ARG, "czk_lanes_copy: range outside the lanes");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    CZK_HIP(ctx, hipMemcpyAsync(dst->p + 4 * (dst_lane * dst->len + dst_elem), src->p + 4 * (src_lane * src->len + src_elem), n * 32,
                                hipMemcpyDeviceToDevice, ctx->stream));
    return CZK_OK;
}

extern "C" int czk_lanes_zero(czk_ctx* ctx, czk_lanes* dst, size_t lane, size_t elem, size_t n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!in_range(dst, lane, elem, n)) return set_err(ctx, CZK_ERR_ARG, "czk_lanes_zero: range outside the lanes");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    CZK_HIP(ctx, hipMemsetAsync(dst->p + 4 * (lane * dst->len + elem), 0, n * 32, ctx->stream));
    return CZK_OK;
}

// ---- czk_fr_copy_3d: one strided copy / fill kernel for every re-layout between transforms -------------------------------------
namespace czk {
struct Copy3 {
    size_t n0, n1, n2, d0, d1, d2, s0, s1, s2;
};
// one thread per Fr element (two 16-byte accesses); i2 is the fastest index, so unit inner strides give coalesced runs
__global__ void k_copy_3d(u64* dst, const u64* src, Copy3 c) {
    const size_t total = c.n0 * c.n1 * c.n2;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t i2 = t % c.n2, r = t / c.n2, i1 = r % c.n1, i0 = r / c.n1;
        ulonglong2* d = reinterpret_cast<ulonglong2*>(dst + 4 * (i0 * c.d0 + i1 * c.d1 + i2 * c.d2));
        if (src) {
            const ulonglong2* s = reinterpret_cast<const ulonglong2*>(src + 4 * (i0 * c.s0 + i1 * c.s1 + i2 * c.s2));
            const ulonglong2 lo = s[0], hi = s[1];
            d[0] = lo;
            d[1] = hi;
        } else {
            d[0] = d[1] = make_ulonglong2(0, 0);
        }
    }
}
}  // namespace czk

extern "C" int czk_fr_copy_3d(czk_ctx* ctx, uint64_t* dst, const size_t* dst_stride, const uint64_t* src, const size_t* src_stride, const size_t* n) {
    if (!ctx) return CZK_ERR_ARG;
    if (!dst_stride || !n || (src && !src_stride)) return set_err(ctx, CZK_ERR_ARG, "czk_fr_copy_3d: null stride / extent");
    const size_t total = n[0] * n[1] * n[2];
    if (!total) return CZK_OK;
    if (!dst) return set_err(ctx, CZK_ERR_ARG, "czk_fr_copy_3d: null destination");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    // contiguous cases go to the copy engine / memset path.  The stride of a dimension of extent 1 is never used, so it does not count (the provers pass
    // stride 0 with n[0] == 1 for every two-dimensional re-layout)
    auto dense = [&](const size_t* st) { return (n[2] == 1 || st[2] == 1) && (n[1] == 1 || st[1] == n[2]) && (n[0] == 1 || st[0] == n[1] * n[2]); };
    const bool dst_dense = dense(dst_stride);
    if (dst_dense && !src) {
        CZK_HIP(ctx, hipMemsetAsync(dst, 0, total * 32, ctx->stream));
        return CZK_OK;
    }
    if (dst_dense && dense(src_stride)) {
        CZK_HIP(ctx, hipMemcpyAsync(dst, src, total * 32, hipMemcpyDeviceToDevice, ctx->stream));
        return CZK_OK;
    }
    Copy3 c{n[0], n[1], n[2], dst_stride[0], dst_stride[1], dst_stride[2], src ? src_stride[0] : 0, src ? src_stride[1] : 0, src ? src_stride[2] : 0};
    size_t blocks = (total + 255) / 256, cap = (size_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_copy_3d, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (u64*)dst, (const u64*)src, c);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

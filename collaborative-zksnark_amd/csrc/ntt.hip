// ntt.hip -- batched in-order radix-2 NTT over BLS12-377 Fr for gfx950, plus the share-local pointwise
// kernels of the Groth16 witness map.
//
// Replaces (value-for-value, every output limb identical):
//   Radix2EvaluationDomain::{fft,ifft,coset_ifft}_in_place   algebra/poly/src/domain/radix2/mod.rs:99-117
//   EvaluationDomain::coset_fft_in_place (trait default)        algebra/poly/src/domain/mod.rs:139-142
//   io_helper / oi_helper / derange                             algebra/poly/src/domain/radix2/fft.rs:140-260
//
// Design (MI355X-first, not the reference's stage-by-stage sweep over DRAM):
//   * log2 D stages are split into ceil(n/7) passes; a pass keeps a (2^K rows x T columns) tile of Fr in
//     LDS (K<=7, T<=16: 64 KiB, 68.5 KiB for the padded last pass at K = 7 -- within gfx950's 160 KiB per CU for two workgroups;
//     checked against the device at launch) and runs its K decimation-in-frequency stages there,
//     so a 2^21 transform touches HBM 3 times instead of 21 (+ a separate bit-reversal sweep).
//   * strided passes read/write T consecutive elements per row (512 B runs); the last pass owns T contiguous
//     2^K chunks whose bit-reversed chunk ids are consecutive, so its transposed store is the bit-reversal
//     permutation AND is written in 512 B runs: natural order in, natural order out, no derange kernel.
//   * coset pre-scale (g^i), zero-extension to D, and the inverse's 1/D (or 1/D * g^-i) post-scale are fused
//     into the first load / last store.
//   * twiddles: one per-stage-compacted table per domain (D-1 entries, like the reference's cache-aligned
//     root compaction fft.rs:194-200) so a wave reads consecutive entries; the last pass stages its 2^K-1
//     twiddles in LDS.
//   * lanes (sh / mac share lanes, several parties) ride on gridDim.y.
// Arithmetic is integer VALU (v_mad_u64_u32); there is no MFMA-shaped work here.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "czk_internal.h"
#include "ntt_pass.h"

namespace czk {

// ------------------------------------------------------------------------------------------------
// LDS helpers: an Fr is two 16-byte slots
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Fr lds_get(const uint4* s, unsigned idx) {
    uint4 a = s[2 * idx], b = s[2 * idx + 1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_put(uint4* s, unsigned idx, const Fr& v) {
    s[2 * idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    s[2 * idx + 1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr gfr_load(const u64* base, size_t idx) { return fp_load<FrParams>(base + 4 * idx); }
__device__ __forceinline__ void gfr_store(u64* base, size_t idx, const Fr& v) { fp_store<FrParams>(base + 4 * idx, v); }

__device__ __forceinline__ unsigned bitrev(unsigned v, unsigned bits) {
    return bits ? (__brev(v) >> (32 - bits)) : 0u;
}

struct PassArgs {
    const u64* in;
    u64* out;
    const u64* tw;        // per-stage compacted twiddles
    const u64* prescale;  // g^i table or null (first pass of a coset fft)
    const u64* posttab;   // size_inv * g^-i table or null (last pass of a coset ifft)
    Fr postconst;         // size_inv (last pass of an ifft)
    int post_mode;        // 0 none, 1 constant, 2 table
    unsigned n;           // log2 D
    unsigned s_lo;        // lowest stage bit of this pass
    unsigned K;           // stages in this pass
    unsigned logT;        // log2 columns
    size_t in_len;        // elements >= in_len read as zero (only honoured when `first`)
    int first;
    size_t lane_stride;   // elements between lanes (= D)
    size_t in_lane_stride;   // ... of `in` in the first pass (out-of-place transforms read the caller's source lanes)
};

// One strided pass: stage bits [s_lo, s_lo + K), s_lo > 0.
__global__ __launch_bounds__(256) void k_ntt_strided(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const unsigned tid = threadIdx.x;
    const unsigned T = 1u << a.logT;
    const unsigned rows = 1u << a.K;
    const unsigned lb_bits = a.s_lo - a.logT;
    const unsigned B = blockIdx.x;
    const unsigned Lb = B & ((1u << lb_bits) - 1u);
    const size_t H = B >> lb_bits;
    const size_t base = (H << (a.s_lo + a.K)) | ((size_t)Lb << a.logT);
    const u64* in = a.in + 4 * (a.first ? a.in_lane_stride : a.lane_stride) * blockIdx.y;
    u64* out = a.out + 4 * a.lane_stride * blockIdx.y;

    for (unsigned e = tid; e < rows * T; e += 256) {
        unsigned r = e >> a.logT, t = e & (T - 1);
        size_t gi = base | ((size_t)r << a.s_lo) | t;
        Fr v;
        if (!a.first || gi < a.in_len) {
            v = gfr_load(in, gi);
            if (a.prescale) v = fp_mul(v, gfr_load(a.prescale, gi));
        } else {
            v = Fr::zero();
        }
        lds_put(smem, e, v);
    }
    __syncthreads();
    const unsigned low0 = Lb << a.logT;
    for (int q = (int)a.K - 1; q >= 0; q--) {
        const unsigned s = a.s_lo + q;
        const u64* tws = a.tw + 4 * (((size_t)1 << s) - 1);
        for (unsigned bf = tid; bf < (rows >> 1) * T; bf += 256) {
            unsigned t = bf & (T - 1), rr = bf >> a.logT;
            unsigned rlow = rr & ((1u << q) - 1u);
            unsigned r0 = ((rr >> q) << (q + 1)) | rlow;
            unsigned r1 = r0 | (1u << q);
            Fr lo = lds_get(smem, r0 * T + t), hi = lds_get(smem, r1 * T + t);
            size_t j = ((size_t)rlow << a.s_lo) | (low0 | t);   // element index mod 2^s
            Fr w = gfr_load(tws, j);
            lds_put(smem, r0 * T + t, fp_add(lo, hi));
            lds_put(smem, r1 * T + t, fp_mul(fp_sub(lo, hi), w));
        }
        __syncthreads();
    }
    for (unsigned e = tid; e < rows * T; e += 256) {
        unsigned r = e >> a.logT, t = e & (T - 1);
        size_t gi = base | ((size_t)r << a.s_lo) | t;
        gfr_store(out, gi, lds_get(smem, e));
    }
}

// Last pass: stage bits [0, K); T contiguous chunks of 2^K; transposed (bit-reversing) store.
__global__ __launch_bounds__(256) void k_ntt_final(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const unsigned tid = threadIdx.x;
    const unsigned T = 1u << a.logT;
    const unsigned c = a.K;
    const unsigned len = 1u << c;
    const unsigned RS = len + 1;   // padded row stride (elements): transposed reads stay <= 2-way conflicted
    uint4* tw_lds = smem + 2 * (size_t)RS * T;
    const unsigned hb = a.n - c;   // bits of the chunk id
    const unsigned B = blockIdx.x;
    const u64* in = a.in + 4 * (a.first ? a.in_lane_stride : a.lane_stride) * blockIdx.y;
    u64* out = a.out + 4 * a.lane_stride * blockIdx.y;

    for (unsigned e = tid; e + 1 < len; e += 256) lds_put(tw_lds, e, gfr_load(a.tw, e));   // stages 0..c-1: 2^c - 1 entries
    for (unsigned e = tid; e < len * T; e += 256) {
        unsigned t = e >> c, l = e & (len - 1);
        size_t h = bitrev(B * T + t, hb);
        size_t gi = (h << c) | l;
        Fr v;
        if (!a.first || gi < a.in_len) {
            v = gfr_load(in, gi);
            if (a.prescale) v = fp_mul(v, gfr_load(a.prescale, gi));
        } else {
            v = Fr::zero();
        }
        lds_put(smem, t * RS + l, v);
    }
    __syncthreads();
    for (int q = (int)c - 1; q >= 0; q--) {
        for (unsigned bf = tid; bf < (len >> 1) * T; bf += 256) {
            unsigned t = bf >> (c - 1), rr = bf & ((len >> 1) - 1);
            unsigned llow = rr & ((1u << q) - 1u);
            unsigned l0 = ((rr >> q) << (q + 1)) | llow;
            unsigned l1 = l0 | (1u << q);
            Fr lo = lds_get(smem, t * RS + l0), hi = lds_get(smem, t * RS + l1);
            lds_put(smem, t * RS + l0, fp_add(lo, hi));
            // the last stage's only twiddle is w^0 = 1 (the reference multiplies by roots[0] = 1, fft.rs:167-180: same value)
            Fr d = fp_sub(lo, hi);
            lds_put(smem, t * RS + l1, q == 0 ? d : fp_mul(d, lds_get(tw_lds, ((1u << q) - 1u) + llow)));
        }
        __syncthreads();
    }
    for (unsigned e = tid; e < len * T; e += 256) {
        unsigned t = e & (T - 1), l = e >> a.logT;
        size_t k = ((size_t)bitrev(l, c) << hb) | (size_t)(B * T + t);
        Fr v = lds_get(smem, t * RS + l);
        if (a.post_mode == 1) v = fp_mul(v, a.postconst);
        else if (a.post_mode == 2) v = fp_mul(v, gfr_load(a.posttab, k));
        gfr_store(out, k, v);
    }
}

// ------------------------------------------------------------------------------------------------
// table generation
// ------------------------------------------------------------------------------------------------
// out[i] = c * base^i for i < count; each thread owns 64 consecutive entries.
__global__ void k_pow_table(u64* out, size_t count, Fr base, Fr c) {
    size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = chunk * 64;
    if (start >= count) return;
    Fr v = fp_mul(c, fp_pow_u64(base, (u64)start));
    size_t end = start + 64 < count ? start + 64 : count;
    for (size_t i = start; i < end; i++) {
        gfr_store(out, i, v);
        v = fp_mul(v, base);
    }
}

struct StageRoots {
    Fr w[48];   // w[s] = root^(2^(n-1-s)) for s < n  (TWO_ADICITY = 47 bounds n)
};
// flat per-stage table: entry (2^s - 1 + j) = w[s]^j, j < 2^s
__global__ void k_twiddle_table(u64* out, unsigned n, const StageRoots* roots) {
    size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = ((size_t)1 << n) - 1;
    size_t start = chunk * 64;
    if (start >= total) return;
    size_t end = start + 64 < total ? start + 64 : total;
    unsigned s_prev = 0xffffffffu;
    Fr v = Fr::one(), w = Fr::one();
    for (size_t i = start; i < end; i++) {
        unsigned s = 63 - __clzll((unsigned long long)(i + 1));
        size_t j = (i + 1) - ((size_t)1 << s);
        if (s != s_prev) {
            w = roots->w[s];
            v = fp_pow_u64(w, (u64)j);
            s_prev = s;
        }
        gfr_store(out, i, v);
        v = fp_mul(v, w);
    }
}

// ------------------------------------------------------------------------------------------------
// pointwise kernels (grid-stride, one Fr per thread-iteration, 2 x 16 B accesses per element)
// ------------------------------------------------------------------------------------------------
__global__ void k_vec_op(int op, const u64* a, const u64* b, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = gfr_load(a, i), y = gfr_load(b, i), r;
        if (op == CZK_OP_ADD) r = fp_add(x, y);
        else if (op == CZK_OP_SUB) r = fp_sub(x, y);
        else r = fp_mul(x, y);
        gfr_store(out, i, r);
    }
}
__global__ void k_vec_scale(const u64* a, Fr k, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        gfr_store(out, i, fp_mul(gfr_load(a, i), k));
}
// the same with the scalar read from DEVICE memory (czk_fr_vec_scale with CZK_MEM_DEVICE): no host round trip, no stream synchronisation
__global__ void k_vec_scale_dev(const u64* a, const u64* kp, u64* out, size_t n) {
    const Fr k = gfr_load(kp, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        gfr_store(out, i, fp_mul(gfr_load(a, i), k));
}
// (ab - c) * k  -- r1cs_to_qap.rs:105-109 fused
__global__ void k_sub_scale(const u64* ab, const u64* c, Fr k, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        gfr_store(out, i, fp_mul(fp_sub(gfr_load(ab, i), gfr_load(c, i)), k));
}
__global__ void k_beaver(const u64* x, const u64* y, const u64* z, const u64* sx, const u64* oy, int add_open, u64* out,
                         size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr vsx = gfr_load(sx, i), voy = gfr_load(oy, i);
        Fr r = fp_sub(gfr_load(z, i), fp_mul(gfr_load(y, i), vsx));
        r = fp_sub(r, fp_mul(gfr_load(x, i), voy));
        if (add_open) r = fp_add(r, fp_mul(vsx, voy));
        gfr_store(out, i, r);
    }
}
__global__ void k_repr(int to_mont, const u64* a, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr v = gfr_load(a, i);
        gfr_store(out, i, to_mont ? fp_from_repr(v) : fp_into_repr(v));
    }
}

// SpdzFieldShare::batch_open, local part (share/spdz.rs:166-185): value = sum of the parties' sh lanes; the MAC
// check sum_p (mac_share_p * value - mac_p) must vanish; non-zero entries are counted
__global__ void k_spdz_open(const u64* shares, size_t parties, size_t n, u64* out_value, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr v = gfr_load(shares, i);                       // party 0, sh lane
        Fr chk = fp_neg(gfr_load(shares + 4 * n, i));     // - mac_0
        for (size_t p = 1; p < parties; p++) {
            v = fp_add(v, gfr_load(shares + 4 * n * (2 * p), i));
            chk = fp_sub(chk, gfr_load(shares + 4 * n * (2 * p + 1), i));
        }
        chk = fp_add(chk, v);                             // + mac_share_0 * value, mac_share_0 = 1 (king)
        gfr_store(out_value, i, v);
        if (!chk.is_zero()) atomicAdd(bad, 1ull);
    }
}

static unsigned grid_for(czk_ctx* ctx, size_t n) {
    size_t blocks = (n + 255) / 256;
    size_t cap = (size_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    return blocks ? (unsigned)blocks : 1u;
}

// ------------------------------------------------------------------------------------------------
// domain object (host side of Radix2EvaluationDomain::new, radix2/mod.rs:51-82)
// ------------------------------------------------------------------------------------------------
static Fr host_root_of_unity(unsigned log_n) {
    // fr.rs:21-28 LARGE_SUBGROUP_ROOT_OF_UNITY (Montgomery), cubed, then squared 47 - log_n times
    Fr large;
    const u32 lr[8] = {0xc790c167u, 0x9bfe9d90u, 0x39013bffu, 0x7175a69eu, 0xadabcf93u, 0x3fbbb698u, 0xd6f0dc97u, 0x0c59f8d8u};
    for (int i = 0; i < 8; i++) large.l[i] = lr[i];
    Fr w = fp_pow_u64(large, 3);
    for (unsigned i = log_n; i < 47; i++) w = fp_sqr(w);
    return w;
}
static Fr host_generator() {
    // fr.rs:69-74 GENERATOR: the Montgomery limbs there encode 22 (22 * R mod r), so build it from 22
    Fr g = Fr::zero();
    g.l[0] = 22;
    return fp_from_repr(g);
}

int get_domain(czk_ctx* ctx, unsigned log_d, DomainTables** out) {
    if (log_d > 47) return set_err(ctx, CZK_ERR_SIZE, "domain larger than 2^TWO_ADICITY (radix2/mod.rs:61-63)");
    auto it = ctx->domains.find(log_d);
    if (it != ctx->domains.end()) {
        *out = &it->second;
        return CZK_OK;
    }
    DomainTables d;
    d.log_d = log_d;
    const u64 D = (u64)1 << log_d;
    d.group_gen = host_root_of_unity(log_d);
    d.group_gen_inv = fp_inv(d.group_gen);
    Fr dsz = Fr::zero();
    dsz.l[0] = (u32)D;
    dsz.l[1] = (u32)(D >> 32);
    d.size_inv = fp_inv(fp_from_repr(dsz));
    d.generator = host_generator();
    d.generator_inv = fp_inv(d.generator);
    d.vanishing_inv = fp_inv(fp_sub(fp_pow_u64(d.generator, D), Fr::one()));
    ctx->domains[log_d] = d;
    *out = &ctx->domains[log_d];
    return CZK_OK;
}

constexpr unsigned NTT2_MIN_LOG = 11;   // domains from 2^11 on run the second-generation passes (pass sizes 5..7)

// device tables are built lazily (a 2^47 domain has valid constants but no tables)
static int ensure_tables(czk_ctx* ctx, DomainTables* d, bool need_coset_fwd, bool need_coset_inv) {
    const unsigned n = d->log_d;
    const size_t D = (size_t)1 << n;
    if ((!d->tw_fwd || !d->tw_inv) && n > 0) {
        StageRoots hr[2];
        for (unsigned s = 0; s < n; s++) {
            Fr wf = d->group_gen, wi = d->group_gen_inv;
            for (unsigned k = 0; k < n - 1 - s; k++) {
                wf = fp_sqr(wf);
                wi = fp_sqr(wi);
            }
            hr[0].w[s] = wf;
            hr[1].w[s] = wi;
        }
        StageRoots* dr = nullptr;
        CZK_HIP(ctx, hipMalloc(&dr, sizeof(hr)));
        CZK_HIP(ctx, hipMemcpyAsync(dr, hr, sizeof(hr), hipMemcpyHostToDevice, ctx->stream));
        if (alloc_table_pair(ctx, &d->tw_fwd, &d->tw_inv, (D - 1 ? D - 1 : 1) * 32) != CZK_OK) {
            (void)hipFree(dr);
            return CZK_ERR_NOMEM;
        }
        size_t chunks = (D - 1 + 63) / 64;
        unsigned blocks = (unsigned)((chunks + 127) / 128);
        hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(128), 0, ctx->stream, d->tw_fwd, n, dr);
        hipLaunchKernelGGL(k_twiddle_table, dim3(blocks), dim3(128), 0, ctx->stream, d->tw_inv, n, dr + 1);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(dr);
        if (e != hipSuccess) return drop_table_pair(ctx, &d->tw_fwd, &d->tw_inv, e);
    }
    size_t chunks = (D + 63) / 64;
    unsigned blocks = (unsigned)((chunks + 127) / 128);
    if (need_coset_fwd && !d->coset_fwd) {
        CZK_HIP(ctx, hipMalloc(&d->coset_fwd, D * 32));
        hipLaunchKernelGGL(k_pow_table, dim3(blocks), dim3(128), 0, ctx->stream, d->coset_fwd, D, d->generator, Fr::one());
        CZK_HIP(ctx, hipGetLastError());
    }
    if (need_coset_inv && !d->coset_inv) {
        CZK_HIP(ctx, hipMalloc(&d->coset_inv, D * 32));
        hipLaunchKernelGGL(k_pow_table, dim3(blocks), dim3(128), 0, ctx->stream, d->coset_inv, D, d->generator_inv, d->size_inv);
        CZK_HIP(ctx, hipGetLastError());
    }
    if (n >= NTT2_MIN_LOG) {   // second-generation passes: the same tables in the unsaturated residue system (fru.h)
        if (!d->twu_fwd || !d->twu_inv) {
            CZK_TRY(alloc_table_pair(ctx, &d->twu_fwd, &d->twu_inv, (D - 1) * 36));
            launch_table_to_u(ctx->stream, d->tw_fwd, D - 1, d->twu_fwd);
            launch_table_to_u(ctx->stream, d->tw_inv, D - 1, d->twu_inv);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return drop_table_pair(ctx, &d->twu_fwd, &d->twu_inv, e);
        }
        if (need_coset_fwd && !d->cosetu_fwd) {
            CZK_HIP(ctx, hipMalloc(&d->cosetu_fwd, D * 36));
            launch_table_to_u(ctx->stream, d->coset_fwd, D, d->cosetu_fwd);
            CZK_HIP(ctx, hipGetLastError());
        }
        if (need_coset_inv && !d->cosetu_inv) {
            CZK_HIP(ctx, hipMalloc(&d->cosetu_inv, D * 36));
            launch_table_to_u(ctx->stream, d->coset_inv, D, d->cosetu_inv);
            CZK_HIP(ctx, hipGetLastError());
        }
    }
    return CZK_OK;
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
int ntt_device(czk_ctx* ctx, u64* data, unsigned log_d, size_t lanes, int kind, size_t in_len, const u64* src, size_t src_stride) {
    if (kind < 0 || kind > 3) return set_err(ctx, CZK_ERR_ARG, "bad ntt kind");
    if (!src) {   // in place
        src = data;
        src_stride = (size_t)1 << log_d;
    }
    DomainTables* d = nullptr;
    CZK_TRY(get_domain(ctx, log_d, &d));
    const size_t D = (size_t)1 << log_d;
    if (in_len > D) return set_err(ctx, CZK_ERR_SIZE, "coeffs.len() > domain size (radix2/mod.rs:100)");
    if (lanes == 0) return CZK_OK;
    if (log_d > 30) return set_err(ctx, CZK_ERR_SIZE, "domain above 2^30 exceeds this build's table budget");
    const bool inverse = (kind == CZK_IFFT || kind == CZK_COSET_IFFT);
    CZK_TRY(ensure_tables(ctx, d, kind == CZK_COSET_FFT, kind == CZK_COSET_IFFT));

    const unsigned n = log_d;
    // split the n stages into ceil(n/7) near-equal groups, highest bits first
    unsigned m = n == 0 ? 1 : (n + 6) / 7;
    unsigned groups[8];
    for (unsigned i = 0; i < m; i++) groups[i] = n / m + (i < n % m ? 1 : 0);

    PassArgs a;
    a.tw = inverse ? d->tw_inv : d->tw_fwd;
    a.n = n;
    a.in_len = in_len;
    a.lane_stride = D;
    a.in_lane_stride = src_stride;
    a.postconst = d->size_inv;

    u64* scratch = nullptr;
    if (m > 1) {
        CZK_TRY(ensure_buf(ctx, ctx->ntt_scratch, lanes * D * (n >= NTT2_MIN_LOG ? NTT2_SCRATCH_ELEM_BYTES : 32)));
        scratch = (u64*)ctx->ntt_scratch.p;
    }
    unsigned s_hi_plus1 = n;
    if (n >= NTT2_MIN_LOG && !ctx->ntt_gen1) {
        // second-generation passes (ntt_pass.hip): register-resident radix-8 / radix-4 groups, unsaturated arithmetic
        Pass2Args b;
        b.tw = inverse ? d->twu_inv : d->twu_fwd;
        b.n = n;
        b.in_len = in_len;
        b.lane_stride = D;
        b.in_lane_stride = src_stride;
        b.postconst = host_fr_to_u(d->size_inv);
        for (unsigned p = 0; p < m; p++) {
            const bool first = (p == 0), last = (p == m - 1);
            const unsigned K = groups[p];
            b.s_lo = s_hi_plus1 - K;
            s_hi_plus1 = b.s_lo;
            b.first = first ? 1 : 0;
            b.prescale = (first && kind == CZK_COSET_FFT) ? d->cosetu_fwd : nullptr;
            b.posttab = (last && kind == CZK_COSET_IFFT) ? d->cosetu_inv : nullptr;
            b.post_mode = !last ? 0 : (kind == CZK_IFFT ? 1 : (kind == CZK_COSET_IFFT ? 2 : 0));
            b.in = first ? src : scratch;
            b.out = last ? data : scratch;
#ifdef CZK_LAB
            // TIMING EXPERIMENT ONLY (wrong results): the first pass of every coset FFT is not launched -- an upper bound on what fusing it into the last pass
            // of the inverse transform in front of it (witness_map: ifft -> coset_fft on a, b, c) could save: the fused kernel would still run its seven stages
            if (ctx->ntt_skip_coset_first && first && !last && kind == CZK_COSET_FFT) continue;
#endif
            ProfScope ps(ctx, "ntt_pass");
            CZK_TRY(launch_ntt2_pass(ctx, b, K, last, lanes));
        }
        return CZK_OK;
    }
    for (unsigned p = 0; p < m; p++) {
        const bool first = (p == 0), last = (p == m - 1);
        a.K = groups[p];
        a.s_lo = s_hi_plus1 - a.K;
        s_hi_plus1 = a.s_lo;
        a.first = first ? 1 : 0;
        a.prescale = (first && kind == CZK_COSET_FFT) ? d->coset_fwd : nullptr;
        a.posttab = nullptr;
        a.post_mode = 0;
        a.in = first ? src : scratch;
        a.out = last ? data : scratch;
        ProfScope ps(ctx, "ntt_pass");
        if (!last) {
            a.logT = a.s_lo < 4 ? a.s_lo : 4;
            unsigned blocks = (unsigned)(D >> (a.K + a.logT));
            size_t lds = ((size_t)1 << (a.K + a.logT)) * 32;
            hipLaunchKernelGGL(k_ntt_strided, dim3(blocks, (unsigned)lanes), dim3(256), lds, ctx->stream, a);
        } else {
            unsigned hb = n - a.K;
            a.logT = hb < 4 ? hb : 4;
            if (kind == CZK_IFFT) a.post_mode = 1;
            if (kind == CZK_COSET_IFFT) {
                a.post_mode = 2;
                a.posttab = d->coset_inv;
            }
            unsigned blocks = (unsigned)(D >> (a.K + a.logT));
            size_t lds = ((((size_t)1 << a.K) + 1) << a.logT) * 32 + ((size_t)1 << a.K) * 32;
            if (lds > ctx->lds_per_block) return set_err(ctx, CZK_ERR_HIP, "NTT last pass needs more LDS per workgroup than this device offers");
            hipLaunchKernelGGL(k_ntt_final, dim3(blocks, (unsigned)lanes), dim3(256), lds, ctx->stream, a);
        }
        CZK_HIP(ctx, hipGetLastError());
    }
    return CZK_OK;
}

}  // namespace czk

using namespace czk;

// ------------------------------------------------------------------------------------------------
// C ABI (NTT + pointwise part)
// ------------------------------------------------------------------------------------------------
// Host-memory callers with several large lanes (the reference's Vec<MpcField> repacked into share lanes): lane k + 1 goes up
// while lane k comes down -- PCIe is full duplex, but a copy from / to pageable memory blocks its calling thread, so the
// downloads run on a second host thread and a second stream.  The transform of a lane is a fraction of its transfer time.
static int ntt_host_lanes_pipelined(czk_ctx* ctx, uint64_t* data, unsigned log_d, size_t lanes, int kind, size_t in_len) {
    const size_t lane_words = (size_t)4 << log_d, lane_bytes = lane_words * 8;
    DeviceBuf buf;
    CZK_TRY(stage_take(ctx, lanes * lane_bytes, &buf));
    std::vector<hipEvent_t> done(lanes, nullptr);
    hipStream_t down = nullptr;
    hipError_t setup = hipStreamCreateWithFlags(&down, hipStreamNonBlocking);
    for (auto& e : done)
        if (setup == hipSuccess) setup = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    // producer -> downloader hand-over: lanes whose transform has been enqueued / a failure flag, under one mutex
    std::mutex mu;
    std::condition_variable cv;
    size_t ready = 0;
    int failed = 0;
    int rc = CZK_OK;
    if (setup != hipSuccess) {
        rc = set_err(ctx, CZK_ERR_HIP, std::string("NTT lane pipeline set-up: ") + hipGetErrorString(setup));
    } else {
        const int device = ctx->device;
        std::thread downloader([&] {
            (void)hipSetDevice(device);
            for (size_t k = 0; k < lanes; k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return ready > k || failed; });   // sleeps; no host core is spun while the GPU works
                    if (failed) return;
                }
                if (hipStreamWaitEvent(down, done[k], 0) != hipSuccess ||
                    hipMemcpyAsync(data + k * lane_words, (char*)buf.p + k * lane_bytes, lane_bytes, hipMemcpyDeviceToHost, down) != hipSuccess ||
                    hipStreamSynchronize(down) != hipSuccess) {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = 2;
                    return;
                }
            }
        });
        for (size_t k = 0; k < lanes && rc == CZK_OK; k++) {
            u64* lane = (u64*)((char*)buf.p + k * lane_bytes);
            // only the first in_len elements are read by the transform (the tail is taken as zero)
            const size_t up = (in_len < ((size_t)1 << log_d) ? in_len : ((size_t)1 << log_d)) * 32;
            if (up && hipMemcpyAsync(lane, data + k * lane_words, up, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                rc = set_err(ctx, CZK_ERR_HIP, "H2D lane");
            if (rc == CZK_OK) rc = ntt_device(ctx, lane, log_d, 1, kind, in_len);
            if (rc == CZK_OK && hipEventRecord(done[k], ctx->stream) != hipSuccess) rc = set_err(ctx, CZK_ERR_HIP, "event record");
            if (rc == CZK_OK) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ready = k + 1;
                }
                cv.notify_one();
            }
        }
        if (rc != CZK_OK) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!failed) failed = 1;
            }
            cv.notify_one();
        }
        downloader.join();
        if (rc == CZK_OK && failed == 2) rc = set_err(ctx, CZK_ERR_HIP, "D2H lane");
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& e : done)
        if (e) (void)hipEventDestroy(e);
    if (down) (void)hipStreamDestroy(down);
    stage_give(ctx, buf);
    return rc;
}

extern "C" int czk_ntt_fr_to(czk_ctx* ctx, const uint64_t* src, size_t src_stride, uint64_t* dst, unsigned log_d, size_t lanes, int kind, size_t in_len,
                             int mem) {
    if (!ctx) return CZK_ERR_ARG;
    if (!dst || (in_len && !src)) return set_err(ctx, CZK_ERR_ARG, "null ntt argument");
    if (mem != CZK_MEM_DEVICE) return set_err(ctx, CZK_ERR_ARG, "czk_ntt_fr_to takes device memory (host callers copy and use czk_ntt_fr)");
    if (log_d > 47) return set_err(ctx, CZK_ERR_SIZE, "domain too large: log2 D exceeds TWO_ADICITY = 47 (radix2/mod.rs:61-63)");
    if (in_len > src_stride && lanes > 1) return set_err(ctx, CZK_ERR_ARG, "in_len exceeds the source lane stride");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    return ntt_device(ctx, (u64*)dst, log_d, lanes, kind, in_len, in_len ? (const u64*)src : (const u64*)dst, in_len ? src_stride : ((size_t)1 << log_d));
}

extern "C" int czk_ntt_fr(czk_ctx* ctx, uint64_t* data, unsigned log_d, size_t lanes, int kind, size_t in_len, int mem) {
    if (!ctx) return CZK_ERR_ARG;
    if (!data && lanes) return set_err(ctx, CZK_ERR_ARG, "null data");
    if (log_d > 47) return set_err(ctx, CZK_ERR_SIZE, "domain larger than 2^TWO_ADICITY (radix2/mod.rs:61-63)");
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = lanes * ((size_t)32 << log_d);
    if (mem == CZK_MEM_HOST && lanes > 1 && log_d >= 16) return ntt_host_lanes_pipelined(ctx, data, log_d, lanes, kind, in_len);
    Staged s{ctx};
    CZK_TRY(s.to_device(data, bytes, mem));
    CZK_TRY(ntt_device(ctx, (u64*)s.dev, log_d, lanes, kind, in_len));
    return s.to_host(data, bytes);
}

extern "C" int czk_domain_constants(czk_ctx* ctx, unsigned log_d, uint64_t* out24) {
    if (!ctx || !out24) return CZK_ERR_ARG;
    DomainTables* d = nullptr;
    CZK_TRY(get_domain(ctx, log_d, &d));
    const Fr* v[6] = {&d->size_inv, &d->group_gen, &d->group_gen_inv, &d->generator, &d->generator_inv, &d->vanishing_inv};
    for (int k = 0; k < 6; k++)
        for (int i = 0; i < 4; i++) out24[4 * k + i] = (u64)v[k]->l[2 * i] | ((u64)v[k]->l[2 * i + 1] << 32);
    return CZK_OK;
}

extern "C" int czk_fr_vec_op(czk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int mem) {
    if (!ctx || (n && (!a || !b || !out)) || op < 0 || op > 2) return ctx ? set_err(ctx, CZK_ERR_ARG, "bad vec_op argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged sa{ctx}, sb{ctx}, so{ctx};
    CZK_TRY(sa.to_device(a, n * 32, mem));
    CZK_TRY(sb.to_device(b, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    hipLaunchKernelGGL(k_vec_op, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, op, (const u64*)sa.dev, (const u64*)sb.dev, (u64*)so.dev, n);
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}

// czk_ctx_reserve: the tables of one radix-2 domain (all four kinds) and the pass scratch for `lanes` lanes, built now instead of by the first transform
int czk::ntt_reserve(czk_ctx* ctx, unsigned log_d, size_t lanes) {
    DomainTables* d = nullptr;
    CZK_TRY(get_domain(ctx, log_d, &d));
    CZK_TRY(ensure_tables(ctx, d, true, true));
    const size_t D = (size_t)1 << log_d;
    // (twice the lanes: the fused pass of an ifft -> coset_fft pair writes a second set of scratch lanes, ifft_coset_fft_device)
    if (log_d > 7 && lanes) CZK_TRY(ensure_buf(ctx, ctx->ntt_scratch, 2 * lanes * D * (log_d >= NTT2_MIN_LOG ? NTT2_SCRATCH_ELEM_BYTES : 32)));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CZK_OK;
}

extern "C" int czk_fr_vec_scale(czk_ctx* ctx, const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n, int mem) {
    if (!ctx || !k || (n && (!a || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "bad vec_scale argument") : CZK_ERR_ARG;
    const bool host_scalar = mem == (CZK_MEM_DEVICE | CZK_MEM_SCALAR_HOST);
    if (host_scalar) mem = CZK_MEM_HOST + 2;   // (neither of the two plain modes below)
    if (!host_scalar && !valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST, CZK_MEM_DEVICE or CZK_MEM_DEVICE | CZK_MEM_SCALAR_HOST");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (host_scalar) {   // device vectors, the scalar by value with the launch
        hipLaunchKernelGGL(k_vec_scale, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)a, host_fr(k), (u64*)out, n);
        CZK_HIP(ctx, hipGetLastError());
        return CZK_OK;
    }
    if (mem == CZK_MEM_DEVICE) {
        // the scalar lives in device memory like the vectors: read it in the kernel (a copy to the host would synchronise the stream -- a
        // full pipeline stall in the middle of a prover's round, which is what this call cost the Plonk / Marlin drivers until round 4)
        hipLaunchKernelGGL(k_vec_scale_dev, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, a, k, out, n);
        CZK_HIP(ctx, hipGetLastError());
        return CZK_OK;
    }
    u64 kh[4];
    for (int i = 0; i < 4; i++) kh[i] = k[i];
    Fr kk;
    for (int i = 0; i < 4; i++) {
        kk.l[2 * i] = (u32)kh[i];
        kk.l[2 * i + 1] = (u32)(kh[i] >> 32);
    }
    Staged sa{ctx}, so{ctx};
    CZK_TRY(sa.to_device(a, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    hipLaunchKernelGGL(k_vec_scale, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)sa.dev, kk, (u64*)so.dev, n);
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}

extern "C" int czk_fr_beaver_combine(czk_ctx* ctx, const uint64_t* x, const uint64_t* y, const uint64_t* z, const uint64_t* sx,
                                     const uint64_t* oy, int add_open, uint64_t* out, size_t n, int mem) {
    if (!ctx || (n && (!x || !y || !z || !sx || !oy || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null beaver argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged s0{ctx}, s1{ctx}, s2{ctx}, s3{ctx}, s4{ctx}, so{ctx};
    CZK_TRY(s0.to_device(x, n * 32, mem));
    CZK_TRY(s1.to_device(y, n * 32, mem));
    CZK_TRY(s2.to_device(z, n * 32, mem));
    CZK_TRY(s3.to_device(sx, n * 32, mem));
    CZK_TRY(s4.to_device(oy, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    hipLaunchKernelGGL(k_beaver, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)s0.dev, (const u64*)s1.dev, (const u64*)s2.dev,
                       (const u64*)s3.dev, (const u64*)s4.dev, add_open, (u64*)so.dev, n);
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}

extern "C" int czk_fr_spdz_open(czk_ctx* ctx, const uint64_t* shares, size_t parties, size_t n, uint64_t* out_value, uint64_t* out_bad) {
    if (!ctx || !out_bad || (n && (!shares || !out_value)) || parties == 0) return ctx ? set_err(ctx, CZK_ERR_ARG, "bad spdz_open argument") : CZK_ERR_ARG;
    *out_bad = 0;
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->open_bad) CZK_HIP(ctx, hipMalloc(&ctx->open_bad, 8));
    unsigned long long* bad = ctx->open_bad;
    CZK_HIP(ctx, hipMemsetAsync(bad, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_spdz_open, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)shares, parties, n, (u64*)out_value, bad);
    CZK_HIP(ctx, hipGetLastError());
    unsigned long long hb = 0;
    CZK_HIP(ctx, hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the count is the call's result: the reference asserts on it right here
    *out_bad = hb;
    return CZK_OK;
}

static int repr_common(czk_ctx* ctx, int to_mont, const uint64_t* a, uint64_t* out, size_t n, int mem) {
    if (!ctx || (n && (!a || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null repr argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged sa{ctx}, so{ctx};
    CZK_TRY(sa.to_device(a, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    hipLaunchKernelGGL(k_repr, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, to_mont, (const u64*)sa.dev, (u64*)so.dev, n);
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}
extern "C" int czk_fr_into_repr(czk_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n, int mem) { return repr_common(ctx, 0, a, out, n, mem); }
extern "C" int czk_fr_from_repr(czk_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n, int mem) { return repr_common(ctx, 1, a, out, n, mem); }

// data <- coset_fft(ifft(data)) per lane: the pair R1CStoQAP::witness_map applies to a, b and c (mpc-snarks/src/groth/r1cs_to_qap.rs:85-89, 102-103).  Where the
// two transforms' pass structures line up (the stage bits split into equal groups: 2^21 = 7 + 7 + 7, also 2^12, 2^14, 2^15, 2^18) the last pass of the
// inverse and the first pass of the forward transform run as ONE kernel (ntt_pass.hip k_ntt2_final_first): five trips through HBM instead of six, same values.
static int ifft_coset_fft_device(czk_ctx* ctx, u64* data, unsigned log_d, size_t lanes, size_t in_len) {
    const unsigned n = log_d;
    const unsigned m = n == 0 ? 1 : (n + 6) / 7;
    const bool fuse = ctx->ntt_fuse_pairs && !ctx->ntt_gen1 && n >= NTT2_MIN_LOG && m >= 2 && n % m == 0;   // equal groups: the two passes own the same tiles
    if (!fuse) {
        CZK_TRY(ntt_device(ctx, data, log_d, lanes, CZK_IFFT, in_len));
        return ntt_device(ctx, data, log_d, lanes, CZK_COSET_FFT, (size_t)1 << log_d);
    }
    DomainTables* d = nullptr;
    CZK_TRY(get_domain(ctx, log_d, &d));
    const size_t D = (size_t)1 << log_d;
    if (in_len > D) return set_err(ctx, CZK_ERR_SIZE, "coeffs.len() > domain size (radix2/mod.rs:100)");
    if (lanes == 0) return CZK_OK;
    CZK_TRY(ensure_tables(ctx, d, true, false));
    const unsigned K = n / m;
    // two sets of scratch lanes: the fused pass reads one and writes the other (a block's inputs and outputs are different element sets)
    CZK_TRY(ensure_buf(ctx, ctx->ntt_scratch, 2 * lanes * D * NTT2_SCRATCH_ELEM_BYTES));
    u64* sa = (u64*)ctx->ntt_scratch.p;
    u64* sb = (u64*)((char*)ctx->ntt_scratch.p + lanes * D * NTT2_SCRATCH_ELEM_BYTES);
    Pass2Args inv{}, fwd{};
    inv.tw = d->twu_inv, fwd.tw = d->twu_fwd;
    inv.n = fwd.n = n;
    inv.in_len = in_len, fwd.in_len = D;
    inv.lane_stride = fwd.lane_stride = inv.in_lane_stride = fwd.in_lane_stride = D;
    inv.postconst = fwd.postconst = host_fr_to_u(d->size_inv);
    inv.prescale = inv.posttab = fwd.posttab = nullptr;
    fwd.prescale = nullptr;
    fwd.post_mode = 0;
    // the inverse transform's passes but the last: data -> scratch A
    unsigned s_hi_plus1 = n;
    for (unsigned p = 0; p + 1 < m; p++) {
        inv.s_lo = s_hi_plus1 - K;
        s_hi_plus1 = inv.s_lo;
        inv.first = p == 0 ? 1 : 0;
        inv.post_mode = 0;
        inv.in = p == 0 ? data : sa;
        inv.out = sa;
        ProfScope ps(ctx, "ntt_pass");
        CZK_TRY(launch_ntt2_pass(ctx, inv, K, false, lanes));
    }
    {   // its last pass (bits [0, K), 1 / D) with the forward transform's first (g^i, bits [n - K, n)): scratch A -> scratch B
        inv.s_lo = 0, inv.first = 0, inv.post_mode = 1, inv.in = sa, inv.out = nullptr;
        fwd.s_lo = n - K, fwd.first = 0, fwd.prescale = d->cosetu_fwd, fwd.in = nullptr, fwd.out = sb;
        ProfScope ps(ctx, "ntt_pass");
        CZK_TRY(launch_ntt2_final_first(ctx, inv, fwd, K, lanes));
    }
    s_hi_plus1 = n - K;
    for (unsigned p = 1; p < m; p++) {
        const bool last = p == m - 1;
        fwd.s_lo = s_hi_plus1 - K;
        s_hi_plus1 = fwd.s_lo;
        fwd.first = 0, fwd.prescale = nullptr;
        fwd.in = sb;
        fwd.out = last ? data : sb;
        ProfScope ps(ctx, "ntt_pass");
        CZK_TRY(launch_ntt2_pass(ctx, fwd, K, last, lanes));
    }
    return CZK_OK;
}

extern "C" int czk_witness_map_pre(czk_ctx* ctx, uint64_t* a, size_t a_len, uint64_t* b, size_t b_len, unsigned log_d, size_t lanes) {
    if (!ctx || !a || !b) return ctx ? set_err(ctx, CZK_ERR_ARG, "null witness_map argument") : CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t D = (size_t)1 << log_d;
    if (a_len > D || b_len > D) return set_err(ctx, CZK_ERR_SIZE, "witness_map: more evaluations than the domain holds");
    // elements [len, D) are the `vec![zero; domain_size]` padding (r1cs_to_qap.rs:66-67): the first pass zero-extends
    CZK_TRY(ifft_coset_fft_device(ctx, a, log_d, lanes, a_len));
    CZK_TRY(ifft_coset_fft_device(ctx, b, log_d, lanes, b_len));
    return CZK_OK;
}

extern "C" int czk_witness_map_post(czk_ctx* ctx, uint64_t* ab, uint64_t* c, size_t c_len, unsigned log_d, size_t lanes) {
    if (!ctx || !ab || !c) return ctx ? set_err(ctx, CZK_ERR_ARG, "null witness_map argument") : CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t D = (size_t)1 << log_d;
    if (c_len > D) return set_err(ctx, CZK_ERR_SIZE, "witness_map: more evaluations than the domain holds");
    DomainTables* d = nullptr;
    CZK_TRY(get_domain(ctx, log_d, &d));
    CZK_TRY(ifft_coset_fft_device(ctx, c, log_d, lanes, c_len));
    size_t n = lanes * D;
    hipLaunchKernelGGL(k_sub_scale, dim3(grid_for(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)ab, (const u64*)c, d->vanishing_inv, (u64*)ab, n);
    CZK_HIP(ctx, hipGetLastError());
    CZK_TRY(ntt_device(ctx, ab, log_d, lanes, CZK_COSET_IFFT, D));
    return CZK_OK;
}

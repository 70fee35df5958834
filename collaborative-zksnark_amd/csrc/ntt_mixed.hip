// ntt_mixed.hip -- MixedRadixEvaluationDomain<Fr>::{fft, ifft, coset_fft, coset_ifft}_in_place for domains of size 3 * 2^k
// (algebra/poly/src/domain/mixed_radix.rs:130-157, 286-404; trait-default coset_fft domain/mod.rs:139-142).  The Plonk prover's
// wire domain has 3 * n_gates elements (mpc-plonk/src/relations/flat.rs:282-300), and GeneralEvaluationDomain picks such a
// domain for polynomial products whenever it is the smaller fit.
//
// The reference merges radix-3 first and radix-2 afterwards (decimation in time over a permuted array).  The values are those of
// the plain DFT  X[i] = sum_j x[j] w^(i j),  w = get_root_of_unity(3 * 2^k) = LARGE_SUBGROUP_ROOT_OF_UNITY^(2^(47-k))
// (algebra/ff/src/fields/mod.rs:337-367), so the GPU is free to factor it the other way round and reuse the radix-2 engine:
//     j = 3 a + b,  i = i0 + M c  (M = 2^k; b, c in {0, 1, 2})
//     X[i0 + M c] = sum_b  zeta^(b c) * w^(i0 b) * Y_b[i0],   Y_b = NTT_M(x[3 a + b]) with root w^3 = get_root_of_unity(M)
//   1. k_mixed_split : de-interleave into 3 lanes of M (coalesced reads; coset pre-scale g^j and zero-extension fused in)
//   2. the radix-2 NTT of ntt.hip / ntt_pass.hip on 3 x lanes lanes
//   3. k_mixed_combine: two twiddle multiplies and a radix-3 butterfly with two constant multiplies per i0
//      (s = u1 + u2, d = u1 - u2:  X0 = u0 + s,  X1,2 = u0 - s / 2 +- d (zeta - zeta^2) / 2), post-scale fused in.
// Every output is the fully reduced field value: limbs equal the reference's.
#include "czk_internal.h"

namespace czk {

__device__ __forceinline__ Fr mfr_load(const u64* base, size_t idx) { return fp_load<FrParams>(base + 4 * idx); }
__device__ __forceinline__ void mfr_store(u64* base, size_t idx, const Fr& v) { fp_store<FrParams>(base + 4 * idx, v); }

// out[(lane * 3 + b) * M + a] = in[lane * N + 3 a + b] (* prescale[3 a + b]); indices >= in_len read as zero
__global__ void k_mixed_split(const u64* in, size_t N, size_t M, size_t in_len, const u64* prescale, u64* out) {
    const size_t lane = blockIdx.y;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (size_t)gridDim.x * blockDim.x) {
        Fr v = Fr::zero();
        if (j < in_len) {
            v = mfr_load(in + 4 * lane * N, j);
            if (prescale) v = fp_mul(v, mfr_load(prescale, j));
        }
        mfr_store(out, (lane * 3 + j % 3) * M + j / 3, v);
    }
}

struct MixedConsts {
    Fr half_neg;   // -1/2
    Fr c2;         // (zeta - zeta^2) / 2 for the direction's zeta = w^(+-M)
    Fr post;       // ifft: 1/3 (the radix-2 inverse already carries 1/M); unused otherwise
};
// post_mode: 0 none, 1 constant `post`, 2 table posttab[i] (already contains the 1/3)
__global__ void k_mixed_combine(const u64* y, size_t N, size_t M, const u64* tw, MixedConsts k, int post_mode, const u64* posttab, u64* out) {
    const size_t lane = blockIdx.y;
    const u64* yl = y + 4 * lane * N;
    u64* ol = out + 4 * lane * N;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < M; i0 += (size_t)gridDim.x * blockDim.x) {
        const Fr u0 = mfr_load(yl, i0);
        const Fr u1 = fp_mul(mfr_load(yl, M + i0), mfr_load(tw, i0));            // * w^(i0)
        const Fr u2 = fp_mul(mfr_load(yl, 2 * M + i0), mfr_load(tw, 2 * i0));    // * w^(2 i0)
        const Fr s = fp_add(u1, u2), d = fp_sub(u1, u2);
        const Fr t = fp_add(u0, fp_mul(s, k.half_neg));
        const Fr e = fp_mul(d, k.c2);
        Fr x[3] = {fp_add(u0, s), fp_add(t, e), fp_sub(t, e)};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t i = i0 + (size_t)c * M;
            if (post_mode == 1) x[c] = fp_mul(x[c], k.post);
            else if (post_mode == 2) x[c] = fp_mul(x[c], mfr_load(posttab, i));
            mfr_store(ol, i, x[c]);
        }
    }
}

__global__ void k_pow_table_m(u64* out, size_t count, Fr base, Fr c) {   // out[i] = c * base^i, 64 entries per thread
    size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = chunk * 64;
    if (start >= count) return;
    Fr v = fp_mul(c, fp_pow_u64(base, (u64)start));
    size_t end = start + 64 < count ? start + 64 : count;
    for (size_t i = start; i < end; i++) {
        mfr_store(out, i, v);
        v = fp_mul(v, base);
    }
}

static Fr fr_from_small(u64 v) {
    Fr x = Fr::zero();
    x.l[0] = (u32)v;
    x.l[1] = (u32)(v >> 32);
    return fp_from_repr(x);
}

// domain of size 3 * 2^k: constants as MixedRadixEvaluationDomain::new computes them (mixed_radix.rs:57-107)
int get_mixed_domain(czk_ctx* ctx, unsigned k, MixedDomain** out) {
    if (k > 47) return set_err(ctx, CZK_ERR_SIZE, "two-adicity above TWO_ADICITY = 47 (fields/mod.rs:352-358)");
    auto it = ctx->mixed_domains.find(k);
    if (it != ctx->mixed_domains.end()) {
        *out = &it->second;
        return CZK_OK;
    }
    MixedDomain d;
    d.k = k;
    Fr w;   // LARGE_SUBGROUP_ROOT_OF_UNITY (fr.rs:21-28), q_adicity == small_subgroup_base_adicity: not cubed; squared 47 - k times
    const u32 lr[8] = {0xc790c167u, 0x9bfe9d90u, 0x39013bffu, 0x7175a69eu, 0xadabcf93u, 0x3fbbb698u, 0xd6f0dc97u, 0x0c59f8d8u};
    for (int i = 0; i < 8; i++) w.l[i] = lr[i];
    for (unsigned i = k; i < 47; i++) w = fp_sqr(w);
    d.group_gen = w;
    d.group_gen_inv = fp_inv(w);
    const u64 N = (u64)3 << k;
    d.size_inv = fp_inv(fr_from_small(N));
    d.generator = fr_from_small(22);                       // fr.rs:69-74 GENERATOR decodes to 22
    d.generator_inv = fp_inv(d.generator);
    d.vanishing_inv = fp_inv(fp_sub(fp_pow_u64(d.generator, N), Fr::one()));
    const Fr two_inv = fp_inv(fr_from_small(2));
    d.half_neg = fp_neg(two_inv);
    const Fr zeta = fp_pow_u64(w, (u64)1 << k);            // w^M, a primitive cube root of unity
    d.c2_fwd = fp_mul(fp_sub(zeta, fp_sqr(zeta)), two_inv);
    d.c2_inv = fp_neg(d.c2_fwd);                           // zeta^-1 = zeta^2
    d.third = fp_inv(fr_from_small(3));
    ctx->mixed_domains[k] = d;
    *out = &ctx->mixed_domains[k];
    return CZK_OK;
}

static int ensure_mixed_tables(czk_ctx* ctx, MixedDomain* d, bool coset_fwd, bool coset_inv) {
    const size_t N = (size_t)3 << d->k;
    const unsigned blocks = (unsigned)(((N + 63) / 64 + 127) / 128);
    if (!d->tw_fwd || !d->tw_inv) {
        CZK_TRY(alloc_table_pair(ctx, &d->tw_fwd, &d->tw_inv, N * 32));
        hipLaunchKernelGGL(k_pow_table_m, dim3(blocks), dim3(128), 0, ctx->stream, d->tw_fwd, N, d->group_gen, Fr::one());
        hipLaunchKernelGGL(k_pow_table_m, dim3(blocks), dim3(128), 0, ctx->stream, d->tw_inv, N, d->group_gen_inv, Fr::one());
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return drop_table_pair(ctx, &d->tw_fwd, &d->tw_inv, e);
    }
    if (coset_fwd && !d->coset_fwd) {
        CZK_HIP(ctx, hipMalloc(&d->coset_fwd, N * 32));
        hipLaunchKernelGGL(k_pow_table_m, dim3(blocks), dim3(128), 0, ctx->stream, d->coset_fwd, N, d->generator, Fr::one());
        CZK_HIP(ctx, hipGetLastError());
    }
    if (coset_inv && !d->coset_inv) {   // third * g^-i: the radix-2 inverse below contributes the 1/M
        CZK_HIP(ctx, hipMalloc(&d->coset_inv, N * 32));
        hipLaunchKernelGGL(k_pow_table_m, dim3(blocks), dim3(128), 0, ctx->stream, d->coset_inv, N, d->generator_inv, d->third);
        CZK_HIP(ctx, hipGetLastError());
    }
    return CZK_OK;
}

int ntt_mixed_device(czk_ctx* ctx, u64* data, unsigned k, size_t lanes, int kind, size_t in_len) {
    if (kind < 0 || kind > 3) return set_err(ctx, CZK_ERR_ARG, "bad ntt kind");
    MixedDomain* d = nullptr;
    CZK_TRY(get_mixed_domain(ctx, k, &d));
    const size_t M = (size_t)1 << k, N = 3 * M;
    if (in_len > N) return set_err(ctx, CZK_ERR_SIZE, "coeffs.len() > domain size");
    if (!lanes) return CZK_OK;
    if (k > 29) return set_err(ctx, CZK_ERR_SIZE, "mixed domain above 3 * 2^29 exceeds this build's table budget");
    const bool inverse = kind == CZK_IFFT || kind == CZK_COSET_IFFT;
    CZK_TRY(ensure_mixed_tables(ctx, d, kind == CZK_COSET_FFT, kind == CZK_COSET_IFFT));
    CZK_TRY(ensure_buf(ctx, ctx->mixed_scratch, lanes * N * 32));
    u64* scr = (u64*)ctx->mixed_scratch.p;
    size_t blocks = (N + 255) / 256, cap = (size_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    {
        ProfScope ps(ctx, "ntt_mixed");
        hipLaunchKernelGGL(k_mixed_split, dim3((unsigned)blocks, (unsigned)lanes), dim3(256), 0, ctx->stream, data, N, M, in_len,
                           kind == CZK_COSET_FFT ? d->coset_fwd : nullptr, scr);
    }
    CZK_HIP(ctx, hipGetLastError());
    CZK_TRY(ntt_device(ctx, scr, k, 3 * lanes, inverse ? CZK_IFFT : CZK_FFT, M));
    MixedConsts mc{d->half_neg, inverse ? d->c2_inv : d->c2_fwd, d->third};
    size_t cblocks = (M + 255) / 256;
    if (cblocks > cap) cblocks = cap;
    {
        ProfScope ps(ctx, "ntt_mixed");
        hipLaunchKernelGGL(k_mixed_combine, dim3((unsigned)cblocks, (unsigned)lanes), dim3(256), 0, ctx->stream, scr, N, M, inverse ? d->tw_inv : d->tw_fwd, mc,
                           kind == CZK_IFFT ? 1 : (kind == CZK_COSET_IFFT ? 2 : 0), kind == CZK_COSET_IFFT ? d->coset_inv : nullptr, data);
    }
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

}  // namespace czk

using namespace czk;

// size = 2^k or 3 * 2^k; returns the two-adicity in *k and whether the factor 3 is present
static bool split_size(size_t size, unsigned* k, bool* three) {
    if (!size) return false;
    unsigned t = 0;
    while ((size & 1) == 0) {
        size >>= 1;
        t++;
    }
    if (size != 1 && size != 3) return false;
    *k = t;
    *three = size == 3;
    return true;
}

extern "C" int czk_ntt_fr_mixed(czk_ctx* ctx, uint64_t* data, size_t size, size_t lanes, int kind, size_t in_len, int mem) {
    if (!ctx) return CZK_ERR_ARG;
    if (!data && lanes) return set_err(ctx, CZK_ERR_ARG, "null data");
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    unsigned k = 0;
    bool three = false;
    if (!split_size(size, &k, &three) || k > 47)
        return set_err(ctx, CZK_ERR_SIZE, "no mixed-radix domain of that size: 2^a or 3 * 2^a with a <= 47 (mixed_radix.rs:57-107)");
    if (!three) return czk_ntt_fr(ctx, data, k, lanes, kind, in_len, mem);   // q_adicity = 0: the same root and transform as Radix2EvaluationDomain
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = lanes * size * 32;
    Staged s{ctx};
    CZK_TRY(s.to_device(data, bytes, mem));
    CZK_TRY(ntt_mixed_device(ctx, (u64*)s.dev, k, lanes, kind, in_len));
    return s.to_host(data, bytes);
}

extern "C" int czk_mixed_domain_constants(czk_ctx* ctx, size_t size, uint64_t* out24) {
    if (!ctx || !out24) return CZK_ERR_ARG;
    unsigned k = 0;
    bool three = false;
    if (!split_size(size, &k, &three) || k > 47) return set_err(ctx, CZK_ERR_SIZE, "no mixed-radix domain of that size");
    if (!three) return czk_domain_constants(ctx, k, out24);
    MixedDomain* d = nullptr;
    CZK_TRY(get_mixed_domain(ctx, k, &d));
    const Fr* v[6] = {&d->size_inv, &d->group_gen, &d->group_gen_inv, &d->generator, &d->generator_inv, &d->vanishing_inv};
    for (int j = 0; j < 6; j++)
        for (int i = 0; i < 4; i++) out24[4 * j + i] = (u64)v[j]->l[2 * i] | ((u64)v[j]->l[2 * i + 1] << 32);
    return CZK_OK;
}

extern "C" int czk_fr_powers(czk_ctx* ctx, const uint64_t* g, const uint64_t* c, size_t n, uint64_t* out, int mem) {
    if (!ctx || !g || (n && !out)) return ctx ? set_err(ctx, CZK_ERR_ARG, "null powers argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged so{ctx};
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    hipLaunchKernelGGL(k_pow_table_m, dim3((unsigned)(((n + 63) / 64 + 127) / 128)), dim3(128), 0, ctx->stream, (u64*)so.dev, n, host_fr(g), c ? host_fr(c) : Fr::one());
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}

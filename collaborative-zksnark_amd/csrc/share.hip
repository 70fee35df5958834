// share.hip -- the share-local arithmetic of the reference's `open` protocols, on device buffers ("next" rows f1 / f4 of the
// scope table).  Everything here is pointwise over a vector of Fr elements once the parties' contributions are on this
// GPU (parties as lanes of one GPU, or after the all-gather / gather that plays mpc-net's broadcast / send_to_king):
//   * SPDZ batch_open, round by round as the reference runs it (mpc-algebra/src/share/spdz.rs:166-185):
//       values = sum of the broadcast `sh` lanes                          -> czk_fr_lanes_sum
//       dx_t   = mac_share * value - mac (what goes into atomic_broadcast) -> czk_fr_spdz_dx
//       assert(sum of the broadcast dx_t == 0)                            -> czk_fr_lanes_sum (counts non-zero sums)
//     (czk_fr_spdz_open in ntt.hip is the one-kernel form for the case where all parties' lanes share this GPU.)
//   * GSZ / Shamir batch_open (mpc-algebra/src/share/gsz20/mod.rs:286-300, 434-466): party j holds p(w^j) for the order-n
//     root w of MixedRadixEvaluationDomain::new(n_parties) (algebra/poly/src/domain/mixed_radix.rs:57-107, root rule
//     algebra/ff/src/fields/mod.rs:337-386 with SMALL_SUBGROUP_BASE = 3); open = size-n inverse DFT of the broadcast values,
//     degree check, p(0).                                                  -> czk_fr_gsz_open
#include "czk_internal.h"

namespace czk {

__device__ __forceinline__ Fr sfr_load(const u64* base, size_t idx) { return fp_load<FrParams>(base + 4 * idx); }
__device__ __forceinline__ void sfr_store(u64* base, size_t idx, const Fr& v) { fp_store<FrParams>(base + 4 * idx, v); }

__global__ void k_lanes_sum(const u64* x, size_t k, size_t n, u64* out, unsigned long long* nonzero) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = sfr_load(x, i);
        for (size_t j = 1; j < k; j++) s = fp_add(s, sfr_load(x + 4 * n * j, i));
        if (out) sfr_store(out, i, s);
        if (nonzero && !s.is_zero()) atomicAdd(nonzero, 1ull);
    }
}
__global__ void k_spdz_dx(const u64* value, const u64* mac, Fr mac_share, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        sfr_store(out, i, fp_sub(fp_mul(mac_share, sfr_load(value, i)), sfr_load(mac, i)));
}
// coefficient k of the interpolating polynomial = sum_j v_j * wtab[j * parties + k], wtab = size_inv * w^(-j k)
__global__ void k_gsz_open(const u64* shares, size_t parties, size_t n, const u64* wtab, const u32* degrees, unsigned degree, u64* out_value,
                           unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned d = degrees ? degrees[i] : degree;
        Fr c0 = Fr::zero();
        bool high = false;
        for (size_t k = 0; k < parties; k++) {
            if (k != 0 && k <= d) continue;               // only p(0) and the coefficients above the degree bound matter
            Fr c = Fr::zero();
            for (size_t j = 0; j < parties; j++) c = fp_add(c, fp_mul(sfr_load(shares + 4 * n * j, i), sfr_load(wtab, j * parties + k)));
            if (k == 0) c0 = c;
            else if (!c.is_zero()) high = true;
        }
        sfr_store(out_value, i, c0);                       // p.evaluate(0) = the constant coefficient
        if (high) atomicAdd(bad, 1ull);                    // the reference asserts p.degree() <= d
    }
}

static unsigned share_grid(czk_ctx* ctx, size_t n) {
    size_t blocks = (n + 255) / 256, cap = (size_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    return blocks ? (unsigned)blocks : 1u;
}

// k_adicity (algebra/ff/src/fields/utils.rs:3-14)
static unsigned k_adicity(size_t k, size_t n) {
    unsigned r = 0;
    while (n > 1) {
        if (n % k) return r;
        r++;
        n /= k;
    }
    return r;
}
// F::get_root_of_unity(n) for Fr, LARGE_SUBGROUP branch (fields/mod.rs:337-367): n = 2^a * 3^b with a <= 47, b <= 1
static bool share_root_of_unity(size_t n, Fr* out) {
    const unsigned q_adicity = k_adicity(3, n);
    size_t q_part = 1;
    for (unsigned i = 0; i < q_adicity; i++) q_part *= 3;
    const unsigned two_adicity = k_adicity(2, n);
    if (n == 0 || two_adicity > 47 || q_adicity > 1 || n != ((size_t)1 << two_adicity) * q_part) return false;
    Fr w;
    const u32 lr[8] = {0xc790c167u, 0x9bfe9d90u, 0x39013bffu, 0x7175a69eu, 0xadabcf93u, 0x3fbbb698u, 0xd6f0dc97u, 0x0c59f8d8u};   // fr.rs:21-28
    for (int i = 0; i < 8; i++) w.l[i] = lr[i];
    for (unsigned i = q_adicity; i < 1; i++) w = fp_pow_u64(w, 3);
    for (unsigned i = two_adicity; i < 47; i++) w = fp_sqr(w);
    *out = w;
    return true;
}

}  // namespace czk

using namespace czk;

static int count_readback(czk_ctx* ctx, uint64_t* out_count) {
    unsigned long long hb = 0;
    CZK_HIP(ctx, hipMemcpyAsync(&hb, ctx->open_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the count is the call's result (the reference asserts on it right here)
    *out_count = hb;
    return CZK_OK;
}

extern "C" int czk_fr_lanes_sum(czk_ctx* ctx, const uint64_t* x, size_t k, size_t n, uint64_t* out, uint64_t* out_nonzero) {
    if (!ctx || k == 0 || (n && !x) || (!out && !out_nonzero)) return ctx ? set_err(ctx, CZK_ERR_ARG, "bad lanes_sum argument") : CZK_ERR_ARG;
    if (out_nonzero) *out_nonzero = 0;
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (out_nonzero) {
        if (!ctx->open_bad) CZK_HIP(ctx, hipMalloc(&ctx->open_bad, 8));
        CZK_HIP(ctx, hipMemsetAsync(ctx->open_bad, 0, 8, ctx->stream));
    }
    hipLaunchKernelGGL(k_lanes_sum, dim3(share_grid(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)x, k, n, (u64*)out, out_nonzero ? ctx->open_bad : nullptr);
    CZK_HIP(ctx, hipGetLastError());
    return out_nonzero ? count_readback(ctx, out_nonzero) : CZK_OK;
}

extern "C" int czk_fr_spdz_dx(czk_ctx* ctx, const uint64_t* value, const uint64_t* mac, const uint64_t* mac_share, uint64_t* out, size_t n) {
    if (!ctx || !mac_share || (n && (!value || !mac || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null spdz_dx argument") : CZK_ERR_ARG;
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_spdz_dx, dim3(share_grid(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)value, (const u64*)mac, host_fr(mac_share), (u64*)out, n);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

extern "C" int czk_share_domain_constants(czk_ctx* ctx, size_t parties, uint64_t* out12) {
    if (!ctx || !out12) return ctx ? set_err(ctx, CZK_ERR_ARG, "null share_domain argument") : CZK_ERR_ARG;
    Fr w;
    if (!share_root_of_unity(parties, &w))
        return set_err(ctx, CZK_ERR_SIZE, "no multiplicative subgroup of that order: n_parties must be 2^a or 3 * 2^a (mixed_radix.rs:57-107, gsz20/mod.rs:100-104)");
    Fr sz = Fr::zero();
    sz.l[0] = (u32)parties;
    sz.l[1] = (u32)((u64)parties >> 32);
    const Fr v[3] = {fp_inv(fp_from_repr(sz)), w, fp_inv(w)};
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 4; i++) out12[4 * k + i] = (u64)v[k].l[2 * i] | ((u64)v[k].l[2 * i + 1] << 32);
    return CZK_OK;
}

extern "C" int czk_fr_gsz_open(czk_ctx* ctx, const uint64_t* shares, size_t parties, size_t n, const uint32_t* degrees, unsigned degree,
                               uint64_t* out_value, uint64_t* out_bad) {
    if (!ctx || !out_bad || parties == 0 || (n && (!shares || !out_value))) return ctx ? set_err(ctx, CZK_ERR_ARG, "bad gsz_open argument") : CZK_ERR_ARG;
    *out_bad = 0;
    if (parties > 96) return set_err(ctx, CZK_ERR_SIZE, "gsz_open: more than 96 parties");
    u64 dom[12];
    CZK_TRY(czk_share_domain_constants(ctx, parties, dom));
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    // size_inv * w^(-j k) for j, k < parties (host arithmetic: at most 96^2 multiplies)
    const Fr size_inv = host_fr(dom), winv = host_fr(dom + 8);
    std::vector<u64> tab(4 * parties * parties);
    Fr wj = Fr::one();                                    // w^-j
    for (size_t j = 0; j < parties; j++) {
        Fr e = size_inv;                                  // size_inv * w^(-j k)
        for (size_t k = 0; k < parties; k++) {
            for (int i = 0; i < 4; i++) tab[4 * (j * parties + k) + i] = (u64)e.l[2 * i] | ((u64)e.l[2 * i + 1] << 32);
            e = fp_mul(e, wj);
        }
        wj = fp_mul(wj, winv);
    }
    CZK_TRY(ensure_buf(ctx, ctx->share_tab, tab.size() * 8));
    CZK_HIP(ctx, hipMemcpyAsync(ctx->share_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->open_bad) CZK_HIP(ctx, hipMalloc(&ctx->open_bad, 8));
    CZK_HIP(ctx, hipMemsetAsync(ctx->open_bad, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_gsz_open, dim3(share_grid(ctx, n)), dim3(256), 0, ctx->stream, (const u64*)shares, parties, n, (const u64*)ctx->share_tab.p, degrees, degree,
                       (u64*)out_value, ctx->open_bad);
    CZK_HIP(ctx, hipGetLastError());
    return count_readback(ctx, out_bad);   // also orders the staging vector's lifetime: the copy has completed
}

// core.hip -- context lifecycle, error reporting, workspace management and the O(1) host-side group
// helpers of libczk_hip.so.
#include <string.h>
#include "czk_internal.h"

#include <stdlib.h>
#include <unistd.h>

namespace {
// The MSM pipeline runs on three internal streams next to the caller's stream.  HIP multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4); two streams sharing a queue serialise (measured: accumulate of
// MSM k+1 waited for the reduce of MSM k, -11 % throughput).  Ask for 8 queues unless the user chose otherwise;
// this must happen before the HIP runtime initialises, hence a load-time constructor.
struct HwQueueDefault {
    HwQueueDefault() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} g_hw_queue_default;
}  // namespace

namespace czk {

int set_err(czk_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

int ensure_buf(czk_ctx* ctx, DeviceBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return CZK_OK;
    if (b.p) {
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        CZK_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return set_err(ctx, CZK_ERR_NOMEM, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
    }
    b.bytes = bytes;
    return CZK_OK;
}

int stage_take(czk_ctx* ctx, size_t bytes, DeviceBuf* out) {
    if (!bytes) bytes = 1;
    int best = -1;
    for (size_t i = 0; i < ctx->stage_pool.size(); i++)
        if (ctx->stage_pool[i].bytes >= bytes && (best < 0 || ctx->stage_pool[i].bytes < ctx->stage_pool[best].bytes)) best = (int)i;
    if (best >= 0 && ctx->stage_pool[best].bytes <= 2 * bytes + (1 << 20)) {   // do not pin a huge buffer under a small request
        *out = ctx->stage_pool[best];
        ctx->stage_pool.erase(ctx->stage_pool.begin() + best);
        return CZK_OK;
    }
    out->p = nullptr;
    out->bytes = bytes;
    if (hipMalloc(&out->p, bytes) != hipSuccess) {
        // memory pressure: drop the idle buffers and try once more
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& b : ctx->stage_pool) (void)hipFree(b.p);
        ctx->stage_pool.clear();
        CZK_HIP(ctx, hipMalloc(&out->p, bytes));
    }
    return CZK_OK;
}
void stage_give(czk_ctx* ctx, const DeviceBuf& b) {
    if (!b.p) return;
    constexpr size_t MAX_IDLE = 6;
    if (ctx->stage_pool.size() >= MAX_IDLE) {   // keep the larger ones: evict the smallest
        size_t small = 0;
        for (size_t i = 1; i < ctx->stage_pool.size(); i++)
            if (ctx->stage_pool[i].bytes < ctx->stage_pool[small].bytes) small = i;
        if (ctx->stage_pool[small].bytes >= b.bytes) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(b.p);
            return;
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->stage_pool[small].p);
        ctx->stage_pool.erase(ctx->stage_pool.begin() + small);
    }
    ctx->stage_pool.push_back(b);
}

int Staged::to_device(const void* host, size_t bytes, int mem) {
    if (mem == CZK_MEM_DEVICE) {
        dev = const_cast<void*>(host);
        return CZK_OK;
    }
    DeviceBuf b;
    CZK_TRY(stage_take(ctx, bytes, &b));
    dev = b.p;
    cap = b.bytes;
    owned = true;
    if (host) CZK_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CZK_OK;
}
int Staged::to_host(void* host, size_t bytes) {
    if (!owned) return CZK_OK;
    CZK_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CZK_OK;
}
Staged::~Staged() {
    if (owned && dev) stage_give(ctx, DeviceBuf{dev, cap});
}

static hipEvent_t take_event(czk_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
#ifdef CZK_LAB
// ---- schedule perturbation (option "chaos") -----------------------------------------------------------------------------------------
__global__ void k_chaos_spin(unsigned long long ticks) {   // one wave that holds its stream for `ticks` of the 100 MHz wall clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
unsigned chaos_rand(czk_ctx* ctx) {   // xorshift64*
    unsigned long long x = ctx->chaos;
    x ^= x >> 12, x ^= x << 25, x ^= x >> 27;
    ctx->chaos = x ? x : 0x9E3779B97F4A7C15ull;
    return (unsigned)((x * 0x2545F4914F6CDD1Dull) >> 33);
}
void chaos_point(czk_ctx* ctx, hipStream_t st) {
    if (!ctx->chaos) return;
    const unsigned r = chaos_rand(ctx), how = r & 7;
    if (how >= 3 && how != 6) hipLaunchKernelGGL(k_chaos_spin, dim3(1), dim3(64), 0, st, (unsigned long long)((r >> 3) % 40000));   // up to 400 us on the stream
    if (how >= 6) usleep((r >> 8) % 300);                                                                                          // up to 300 us on the host
}
#endif
ProfScope::ProfScope(czk_ctx* c, const char* n, hipStream_t s) : ctx(c), name(n), st(s ? s : c->stream) {
    chaos_point(ctx, st);
    if (!ctx->profiling) return;
    e0 = take_event(ctx);
    e1 = take_event(ctx);
    (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (e0) {
        (void)hipEventRecord(e1, st);
        ctx->prof[name].pending.emplace_back(e0, e1);
    }
    chaos_point(ctx, st);
}
static void prof_resolve(czk_ctx* ctx) {
    for (auto& kv : ctx->prof) {
        for (auto& pr : kv.second.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.launches++;
                float t0 = 0;
                if (ctx->prof_base && kv.second.intervals.size() < (1u << 20) && hipEventElapsedTime(&t0, ctx->prof_base, pr.first) == hipSuccess)
                    kv.second.intervals.emplace_back(t0, t0 + ms);
            }
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

template <class F>
static void limbs_to_field(const u64* p, F& out);
template <>
void limbs_to_field<Fq>(const u64* p, Fq& out) {
    for (int i = 0; i < 6; i++) {
        out.l[2 * i] = (u32)p[i];
        out.l[2 * i + 1] = (u32)(p[i] >> 32);
    }
}
template <>
void limbs_to_field<Fq2>(const u64* p, Fq2& out) {
    limbs_to_field<Fq>(p, out.c0);
    limbs_to_field<Fq>(p + 6, out.c1);
}
static void field_to_limbs(const Fq& a, u64* p) {
    for (int i = 0; i < 6; i++) p[i] = (u64)a.l[2 * i] | ((u64)a.l[2 * i + 1] << 32);
}
static void field_to_limbs(const Fq2& a, u64* p) {
    field_to_limbs(a.c0, p);
    field_to_limbs(a.c1, p + 6);
}

template <class F>
static void host_jac_to_affine(const u64* jac, size_t n, u64* out_aff, uint8_t* out_inf) {
    // From<Projective> for Affine (short_weierstrass_jacobian.rs:768-789) per point, with ONE field inversion for the whole array (Montgomery's trick, the
    // reference's own batch_inversion, ff/src/fields/mod.rs:704-727): the inverse of an element is unique, so the affine coordinates are the per-point ones.
    // A prover settles a dozen commitments at a transcript point; the Fermat inversion of this host code costs ~0.2 ms each.
    constexpr int W = FieldIO<F>::W64;
    std::vector<Jac<F>> pts(n);
    std::vector<F> pre(n);          // pre[i] = product of the non-zero z before point i
    F acc = F::one();
    for (size_t i = 0; i < n; i++) {
        limbs_to_field<F>(jac + 3 * W * i, pts[i].x);
        limbs_to_field<F>(jac + 3 * W * i + W, pts[i].y);
        limbs_to_field<F>(jac + 3 * W * i + 2 * W, pts[i].z);
        pre[i] = acc;
        if (!pts[i].is_zero()) acc = f_mul(acc, pts[i].z);
    }
    F inv = f_inv(acc);             // 1 / (z_0 z_1 ... ), infinity points left out
    for (size_t i = n; i-- > 0;) {
        Affine<F> a;
        const bool inf = pts[i].is_zero();
        if (inf) {
            a.x = F::zero();
            a.y = F::one();
        } else {
            const F zi = f_mul(inv, pre[i]);
            inv = f_mul(inv, pts[i].z);
            const F zi2 = f_sqr(zi);
            a.x = f_mul(pts[i].x, zi2);
            a.y = f_mul(pts[i].y, f_mul(zi2, zi));
        }
        field_to_limbs(a.x, out_aff + 2 * W * i);
        field_to_limbs(a.y, out_aff + 2 * W * i + W);
        if (out_inf) out_inf[i] = inf ? 1 : 0;
    }
}

template <class F>
static void load_jac(const u64* p, Jac<F>& j) {
    constexpr int W = FieldIO<F>::W64;
    limbs_to_field<F>(p, j.x);
    limbs_to_field<F>(p + W, j.y);
    limbs_to_field<F>(p + 2 * W, j.z);
}
template <class F>
static void store_jac(const Jac<F>& j, u64* p) {
    constexpr int W = FieldIO<F>::W64;
    field_to_limbs(j.x, p);
    field_to_limbs(j.y, p + W);
    field_to_limbs(j.z, p + 2 * W);
}
template <class F>
static void host_jac_add(const u64* a, const u64* b, u64* out) {
    Jac<F> x, y;
    load_jac<F>(a, x);
    load_jac<F>(b, y);
    store_jac<F>(jac_add(x, y), out);
}
template <class F>
static void combine_windows_impl(const u64* src, unsigned W, unsigned c, size_t lanes, u64* out) {
    constexpr int JW = 3 * FieldIO<F>::W64;
    for (size_t l = 0; l < lanes; l++) {
        Jac<F> acc;
        load_jac<F>(src + (l * W + (W - 1)) * JW, acc);
        const unsigned W_hi = msm_full_windows(c);
        for (unsigned w = W - 1; w-- > 0;) {
            for (unsigned k = 0; k < msm_win_width(c, W_hi, w); k++) acc = jac_double(acc);   // the Horner step of variable_base.rs:92-105
            Jac<F> r;
            load_jac<F>(src + (l * W + w) * JW, r);
            acc = jac_add(acc, r);
        }
        store_jac<F>(acc, out + l * JW);
    }
}
void host_combine_windows(int group, const char* src, unsigned W, unsigned c, size_t lanes, uint64_t* out) {
    if (group == CZK_G1) combine_windows_impl<Fq>((const u64*)src, W, c, lanes, out);
    else combine_windows_impl<Fq2>((const u64*)src, W, c, lanes, out);
}
template <class F>
static void host_jac_add_mixed(const u64* a, const u64* b_aff, bool b_inf, u64* out) {
    constexpr int W = FieldIO<F>::W64;
    Jac<F> x;
    load_jac<F>(a, x);
    Affine<F> q;
    limbs_to_field<F>(b_aff, q.x);
    limbs_to_field<F>(b_aff + W, q.y);
    store_jac<F>(jac_add_mixed(x, q, b_inf), out);
}

}  // namespace czk

using namespace czk;

extern "C" int czk_jac_add(czk_ctx* ctx, int group, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !b || !out) return set_err(ctx, CZK_ERR_ARG, "null jac_add argument");
    if (group == CZK_G1) host_jac_add<Fq>(a, b, out);
    else if (group == CZK_G2) host_jac_add<Fq2>(a, b, out);
    else return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    return CZK_OK;
}
extern "C" int czk_jac_add_mixed(czk_ctx* ctx, int group, const uint64_t* a, const uint64_t* b_aff, int b_inf, uint64_t* out) {
    if (!a || !b_aff || !out) return set_err(ctx, CZK_ERR_ARG, "null jac_add_mixed argument");
    if (group == CZK_G1) host_jac_add_mixed<Fq>(a, b_aff, b_inf != 0, out);
    else if (group == CZK_G2) host_jac_add_mixed<Fq2>(a, b_aff, b_inf != 0, out);
    else return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    return CZK_OK;
}

namespace czk {
template <class F>
static void host_jac_scalar_mul(const u64* a, const Fr& k_canonical, u64* out) {
    Jac<F> base, res = Jac<F>::zero();
    load_jac<F>(a, base);
    int top = 255;
    while (top >= 0 && !((k_canonical.l[top / 32] >> (top % 32)) & 1)) top--;      // BitIteratorBE::without_leading_zeros
    for (int i = top; i >= 0; i--) {
        res = jac_double(res);
        if ((k_canonical.l[i / 32] >> (i % 32)) & 1) res = jac_add(res, base);
    }
    store_jac<F>(res, out);
}
template <class F>
static void host_jac_neg(const u64* a, u64* out) {
    Jac<F> p;
    load_jac<F>(a, p);
    p.y = f_neg(p.y);
    store_jac<F>(p, out);
}
}  // namespace czk

extern "C" int czk_jac_scalar_mul(czk_ctx* ctx, int group, const uint64_t* a, const uint64_t* k, int scalar_form, uint64_t* out) {
    if (!a || !k || !out) return set_err(ctx, CZK_ERR_ARG, "null jac_scalar_mul argument");
    if (scalar_form != CZK_SCALAR_CANONICAL && scalar_form != CZK_SCALAR_MONTGOMERY) return set_err(ctx, CZK_ERR_ARG, "bad scalar_form");
    Fr kk = host_fr(k);
    if (scalar_form == CZK_SCALAR_MONTGOMERY) kk = fp_into_repr(kk);
    if (group == CZK_G1) host_jac_scalar_mul<Fq>(a, kk, out);
    else if (group == CZK_G2) host_jac_scalar_mul<Fq2>(a, kk, out);
    else return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    return CZK_OK;
}
extern "C" int czk_jac_neg(czk_ctx* ctx, int group, const uint64_t* a, uint64_t* out) {
    if (!a || !out) return set_err(ctx, CZK_ERR_ARG, "null jac_neg argument");
    if (group == CZK_G1) host_jac_neg<Fq>(a, out);
    else if (group == CZK_G2) host_jac_neg<Fq2>(a, out);
    else return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    return CZK_OK;
}

extern "C" const char* czk_version(void) { return "czk-mi355x 0.1 (gfx950)"; }

extern "C" int czk_ctx_create(czk_ctx** out, int device, void* hip_stream) {
    if (!out) return CZK_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return CZK_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return CZK_ERR_HIP;
    czk_ctx* c = new czk_ctx();
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return CZK_ERR_HIP;
        }
        c->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->num_cu = prop.multiProcessorCount;
        c->lds_per_block = prop.sharedMemPerBlock;
    }
    #ifdef CZK_LAB
    // lab build only: the measurement tools' CZK_* environment switches, translated into options (the product library reads no environment)
    {
        static const char* const ENV[][2] = {{"CZK_NTT_GEN1", "ntt_gen1"}, {"CZK_SORT_ONEPASS", "msm_sort_onepass"}, {"CZK_REDUCE_SAT", "msm_reduce_sat"},
            {"CZK_REDUCE_SAT_G2", "msm_reduce_sat_g2"}, {"CZK_G2_MODE", "msm_g2_mode"}, {"CZK_MSM_AFFINE", "msm_affine_rounds"}, {"CZK_MSM_SLOTS", "msm_slots"},
            {"CZK_STREAM_PRIO", "msm_stream_priority"}, {"CZK_MSM_SAT", "msm_sat"}, {"CZK_MSM_SAT_G2", "msm_sat_g2"}, {"CZK_MSM_NO_TE", "msm_no_te"},
            {"CZK_MSM_FIXED_C", "msm_fixed_c"}, {"CZK_MSM_C_G1", "msm_window_g1"}, {"CZK_MSM_C_G2", "msm_window_g2"}, {"CZK_CHAOS", "chaos"},
            {"CZK_CHAOS_DROP_WAIT", "chaos_drop_wait"}, {"CZK_NTT_SKIP_COSET_FIRST", "ntt_skip_coset_first"}, {"CZK_LANE_INTERLEAVE", "msm_lane_interleave"}};
        for (auto& e : ENV)
            if (const char* v = getenv(e[0])) (void)czk_ctx_set_option(c, e[1], atol(v) ? atol(v) : (v[0] == '0' ? 0 : 1));
    }
#endif
    *out = c;
    return CZK_OK;
}

extern "C" void czk_ctx_destroy(czk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->domains) {
        DomainTables& d = kv.second;
        if (d.tw_fwd) (void)hipFree(d.tw_fwd);
        if (d.tw_inv) (void)hipFree(d.tw_inv);
        if (d.coset_fwd) (void)hipFree(d.coset_fwd);
        if (d.coset_inv) (void)hipFree(d.coset_inv);
        if (d.twu_fwd) (void)hipFree(d.twu_fwd);
        if (d.twu_inv) (void)hipFree(d.twu_inv);
        if (d.cosetu_fwd) (void)hipFree(d.cosetu_fwd);
        if (d.cosetu_inv) (void)hipFree(d.cosetu_inv);
    }
    for (auto& kv : ctx->mixed_domains) {
        MixedDomain& d = kv.second;
        if (d.tw_fwd) (void)hipFree(d.tw_fwd);
        if (d.tw_inv) (void)hipFree(d.tw_inv);
        if (d.coset_fwd) (void)hipFree(d.coset_fwd);
        if (d.coset_inv) (void)hipFree(d.coset_inv);
    }
    if (ctx->mixed_scratch.p) (void)hipFree(ctx->mixed_scratch.p);
    if (ctx->ntt_scratch.p) (void)hipFree(ctx->ntt_scratch.p);
    if (ctx->poly_scratch.p) (void)hipFree(ctx->poly_scratch.p);
    if (ctx->open_bad) (void)hipFree(ctx->open_bad);
    for (auto& b : ctx->stage_pool) (void)hipFree(b.p);
    ctx->stage_pool.clear();
    if (ctx->share_tab.p) (void)hipFree(ctx->share_tab.p);
    msm_pipeline_destroy(ctx);
    xfer_destroy(ctx);
    prof_resolve(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->prof_base) (void)hipEventDestroy(ctx->prof_base);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---- options ------------------------------------------------------------------------------------------------------------------
// The product library reads NO environment variables: what used to be CZK_* switches are explicit per-context options.  Names of the
// second group select kernels that only the lab build (-DCZK_LAB, libczk_hip_lab.so) contains.
namespace {
struct OptDesc {
    const char* name;
    long lo, hi;
    bool before_pipeline;   // must be set before the context's first MSM (streams are created then)
    void (*set)(czk_ctx*, long);
};
const OptDesc OPTIONS[] = {
    {"msm_slots", 1, czk_ctx::MSM_SLOTS, true, [](czk_ctx* c, long v) { c->msm_slots_in_use = (int)v; }},
    {"msm_stream_priority", 0, 2, true, [](czk_ctx* c, long v) { c->msm_stream_prio = (int)v; }},
    {"msm_lane_interleave", 0, 64, false, [](czk_ctx* c, long v) { c->msm_lane_interleave = (int)v; }},
    {"msm_sort_reuse", 0, 1, false, [](czk_ctx* c, long v) { c->msm_sort_reuse = v != 0; }},
    {"msm_sort_onepass", 0, 1, false, [](czk_ctx* c, long v) { c->msm_sort_onepass = v != 0; }},
    {"msm_fixed_c", 0, 1, false, [](czk_ctx* c, long v) { c->msm_fixed_c = v != 0; }},
    {"msm_window_g1", 0, 22, false, [](czk_ctx* c, long v) { c->msm_c_g1 = (unsigned)v; }},
    {"msm_window_g2", 0, 22, false, [](czk_ctx* c, long v) { c->msm_c_g2 = (unsigned)v; }},
    {"ntt_gen1", 0, 1, false, [](czk_ctx* c, long v) { c->ntt_gen1 = v != 0; }},
    {"ntt_fuse_pairs", 0, 1, false, [](czk_ctx* c, long v) { c->ntt_fuse_pairs = v != 0; }},
    {"net_create_timeout_ms", 0, 3600000, false, [](czk_ctx* c, long v) { c->net_create_timeout_ms = v; }},
#ifdef CZK_LAB
    {"msm_affine_rounds", 0, 3, false, [](czk_ctx* c, long v) { c->msm_affine_rounds = (unsigned)v; }},
    {"msm_reduce_sat", 0, 1, false, [](czk_ctx* c, long v) { c->msm_reduce_sat = v != 0; }},
    {"msm_reduce_sat_g2", 0, 1, false, [](czk_ctx* c, long v) { c->msm_reduce_sat_g2 = v != 0; }},
    {"msm_g2_mode", 0, 2, false, [](czk_ctx* c, long v) { c->msm_g2_mode = (int)v; }},
    {"msm_sat", 0, 1, false, [](czk_ctx* c, long v) { c->msm_sat = v != 0; }},
    {"msm_sat_g2", 0, 1, false, [](czk_ctx* c, long v) { c->msm_sat_g2 = v != 0; }},
    {"msm_no_te", 0, 1, false, [](czk_ctx* c, long v) { c->msm_no_te = v != 0; }},
    // (the seed is mixed with the process id and the context's address: the parties of a party layout and the contexts of several proofs in flight must not
    // draw the same delays)
    {"chaos", 0, 0x7fffffff, false, [](czk_ctx* c, long v) { c->chaos = v ? ((unsigned long long)v * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)getpid() << 32) ^ (unsigned long long)(uintptr_t)c : 0; }},
    {"chaos_drop_wait", 0, 1, false, [](czk_ctx* c, long v) { c->chaos_drop_wait = (int)v; }},
    {"ntt_skip_coset_first", 0, 1, false, [](czk_ctx* c, long v) { c->ntt_skip_coset_first = v != 0; }},
    {"msm_sort_reuse_any_inf", 0, 1, false, [](czk_ctx* c, long v) { c->msm_sort_reuse_any_inf = v != 0; }},

#endif
};
}  // namespace
extern "C" int czk_ctx_set_option(czk_ctx* ctx, const char* name, long value) {
    if (!ctx || !name) return CZK_ERR_ARG;
    for (const OptDesc& o : OPTIONS) {
        if (strcmp(o.name, name) != 0) continue;
        if (value < o.lo || value > o.hi || (!strncmp(name, "msm_window", 10) && value != 0 && value < 8))
            return set_err(ctx, CZK_ERR_ARG, std::string("czk_ctx_set_option: value out of range for ") + name);
        if (o.before_pipeline && ctx->s_sort) return set_err(ctx, CZK_ERR_ARG, std::string("czk_ctx_set_option: ") + name + " must be set before the context's first MSM");
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // options take effect between calls, never under enqueued work
        CZK_TRY(msm_pipeline_sync(ctx));
        o.set(ctx, value);
        return CZK_OK;
    }
#ifdef CZK_LAB
    return set_err(ctx, CZK_ERR_ARG, std::string("czk_ctx_set_option: unknown option \"") + name + "\" (lab build)");
#else
    return set_err(ctx, CZK_ERR_ARG, std::string("czk_ctx_set_option: unknown option \"") + name + "\" (product build: the switches of the rejected variants exist in libczk_hip_lab.so only)");
#endif
}
extern "C" int czk_build_is_lab(void) {
#ifdef CZK_LAB
    return 1;
#else
    return 0;
#endif
}

extern "C" int czk_ctx_reserve(czk_ctx* ctx, unsigned ntt_log_d, size_t ntt_lanes, const czk_bases* bases, size_t n_scalars, size_t msm_lanes) {
    if (!ctx) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (ntt_lanes) CZK_TRY(ntt_reserve(ctx, ntt_log_d, ntt_lanes));
    if (bases && msm_lanes) CZK_TRY(msm_reserve(ctx, bases, n_scalars, msm_lanes));
    return CZK_OK;
}
extern "C" void* czk_ctx_stream(const czk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int czk_ctx_sync(czk_ctx* ctx) {
    if (!ctx) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return msm_pipeline_sync(ctx);   // also delivers the results of czk_msm_async calls
}

extern "C" int czk_profile_enable(czk_ctx* ctx, int on) {
    if (!ctx) return CZK_ERR_ARG;
    ctx->profiling = on != 0;
    return CZK_OK;
}
extern "C" int czk_profile_reset(czk_ctx* ctx) {
    if (!ctx) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CZK_TRY(msm_pipeline_sync(ctx));
    prof_resolve(ctx);
    ctx->prof.clear();
    if (!ctx->prof_base) CZK_HIP(ctx, hipEventCreate(&ctx->prof_base));
    CZK_HIP(ctx, hipEventRecord(ctx->prof_base, ctx->stream));
    CZK_HIP(ctx, hipEventSynchronize(ctx->prof_base));
    return CZK_OK;
}
extern "C" int czk_profile_read(czk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches) {
    if (!ctx || !kernel) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CZK_TRY(msm_pipeline_sync(ctx));
    prof_resolve(ctx);
    auto it = ctx->prof.find(kernel);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return CZK_OK;
}

extern "C" int czk_profile_intervals(czk_ctx* ctx, const char* kernel, double* start_ms, double* stop_ms, size_t cap, size_t* n) {
    if (!ctx || !kernel || !n) return CZK_ERR_ARG;
    CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CZK_TRY(msm_pipeline_sync(ctx));
    prof_resolve(ctx);
    auto it = ctx->prof.find(kernel);
    const size_t have = it == ctx->prof.end() ? 0 : it->second.intervals.size();
    *n = have;
    for (size_t i = 0; i < have && i < cap; i++) {
        if (start_ms) start_ms[i] = it->second.intervals[i].first;
        if (stop_ms) stop_ms[i] = it->second.intervals[i].second;
    }
    return CZK_OK;
}
extern "C" int czk_profile_base_offset(czk_ctx* a, czk_ctx* b, double* ms) {
    if (!a || !b || !ms) return CZK_ERR_ARG;
    if (!a->prof_base || !b->prof_base) return set_err(a, CZK_ERR_ARG, "czk_profile_base_offset: czk_profile_reset has not run on both contexts");
    float d = 0;
    CZK_HIP(a, hipEventElapsedTime(&d, a->prof_base, b->prof_base));
    *ms = d;
    return CZK_OK;
}

extern "C" const char* czk_last_error(const czk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int czk_jac_to_affine(czk_ctx* ctx, int group, const uint64_t* jac, size_t n, uint64_t* out_aff, uint8_t* out_inf) {
    // pure host arithmetic: a context is only needed for error text, so NULL is accepted
    if (n && (!jac || !out_aff)) return set_err(ctx, CZK_ERR_ARG, "null jac_to_affine argument");
    if (group == CZK_G1) host_jac_to_affine<Fq>(jac, n, out_aff, out_inf);
    else if (group == CZK_G2) host_jac_to_affine<Fq2>(jac, n, out_aff, out_inf);
    else return set_err(ctx, CZK_ERR_ARG, "group must be CZK_G1 or CZK_G2");
    return CZK_OK;
}

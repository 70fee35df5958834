// czk_internal.h -- context object and helpers shared by the .hip translation units of libczk_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/czk.h"
#include "curve.h"

namespace czk {

// Per-domain device tables for the radix-2 NTT of size D = 2^log_d (see ntt.hip).
struct DomainTables {
    unsigned log_d = 0;
    u64* tw_fwd = nullptr;     // D-1 entries: stage s (bit s) occupies [2^s - 1, 2^(s+1) - 1): w_s^j, w_s = w^(2^(n-1-s))
    u64* tw_inv = nullptr;     // same for w^-1
    u64* coset_fwd = nullptr;  // D entries: g^i, g = 22
    u64* coset_inv = nullptr;  // D entries: size_inv * g^-i
    // the same four tables in the unsaturated residue system of fru.h (9 x u32 per entry: value * 2^261 mod r), used by the
    // second-generation passes (ntt_pass.hip) for domains of 2^11 and more
    uint32_t *twu_fwd = nullptr, *twu_inv = nullptr, *cosetu_fwd = nullptr, *cosetu_inv = nullptr;
    Fr size_inv, group_gen, group_gen_inv, generator, generator_inv, vanishing_inv;
};

// MixedRadixEvaluationDomain of size 3 * 2^k (ntt_mixed.hip)
struct MixedDomain {
    unsigned k = 0;
    u64 *tw_fwd = nullptr, *tw_inv = nullptr;       // 3 * 2^k entries: w^j, w^-j
    u64 *coset_fwd = nullptr, *coset_inv = nullptr; // g^j;  (1/3) g^-j
    Fr size_inv, group_gen, group_gen_inv, generator, generator_inv, vanishing_inv, half_neg, c2_fwd, c2_inv, third;
};

template <class F>
struct GT;
template <>
struct GT<Fq> {
    static constexpr int AW = 12, JW = 18, FW = 6, XW = 24;   // u64 words: affine, Jacobian, field element, XYZZ
};
template <>
struct GT<Fq2> {
    static constexpr int AW = 24, JW = 36, FW = 12, XW = 48;
};

}  // namespace czk

namespace czk {
struct ProfEntry {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<float, float>> intervals;   // (start, stop) in ms after the context's profile base event (czk_profile_intervals)
};
}  // namespace czk

namespace czk {
struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
// What a slot's sort workspace holds when it was made from CZK_MEM_STABLE scalars: a CZK_MEM_SAME_SCALARS call that matches a slot's key in every field reads that
// slot's sorted entries instead of sorting again.  Cleared by czk_ctx_sync / czk_ctx_wait_mark; overwritten by the slot's next sort.
struct MsmSortKey {
    bool valid = false;
    uint64_t seq = 0;          // the call that made the sort (czk_ctx::msm_seq)
    const void* scalars = nullptr;
    size_t n_scalars = 0, lanes = 0, size = 0, nb = 0;
    int form = 0;
    unsigned c = 0, W = 0;
    const czk_bases* bases = nullptr;
};
struct MsmSlot {
    DeviceBuf ws_sort, ws_red, ws_aff;
    hipEvent_t ev_sorted = nullptr, ev_acc = nullptr, ev_fix = nullptr, ev_red = nullptr;
    bool used = false;
    MsmSortKey key;
};
struct MsmPending {   // a host result on its way: pinned staging -> the caller's buffer at czk_ctx_sync / czk_ctx_wait_mark (MSM results, deferred downloads)
    const char* src;
    void* dst;
    size_t bytes;
    bool on_stream = false;   // copied on the context's stream (czk_lanes_download_deferred) rather than on the reduce stream
    // split-window MSMs (bases registered without tables): `src` holds lanes x split_W per-window results that still have to
    // be combined as sum_w 2^(c w) R_w (host, a few hundred point operations) into `dst` = lanes results
    unsigned split_W = 0, c = 0;
    int group = 0;
    size_t lanes = 0;
};
// czk_ctx_mark: the work enqueued so far = an event on the context's stream, one on the reduce stream (results are copied there, in order) and the
// number of host results enqueued before it
struct CtxMark {
    uint64_t id = 0;
    hipEvent_t ev_stream = nullptr, ev_red = nullptr;
    uint64_t upto = 0;   // absolute index: results [delivered, upto) belong to the mark
};
}  // namespace czk

#ifndef CZK_ACC_INTERLEAVE_DEFAULT
#define CZK_ACC_INTERLEAVE_DEFAULT 4u
#endif
struct czk_ctx {
    std::vector<czk::DeviceBuf> stage_pool;   // idle staging buffers of host-memory callers (core.hip)
    // pinned double buffer of czk_lanes_upload / czk_lanes_download (lanes.hip)
    char* xfer_pinned[2] = {nullptr, nullptr};
    hipEvent_t xfer_ev[2] = {nullptr, nullptr};
    bool xfer_busy[2] = {false, false};
    // MSM pipeline (msm.hip)
    hipStream_t s_sort = nullptr, s_acc = nullptr, s_red = nullptr;
    hipEvent_t ev_in = nullptr;
    static constexpr int MSM_SLOTS = 4;   // workspace ring: accumulate(k + MSM_SLOTS) waits for reduce(k)
    czk::MsmSlot msm_slots[MSM_SLOTS];
    int msm_next_slot = 0;
    // CZK_MEM_SAME_SCALARS names the scalars of the most recent call made WITHOUT the flag (the leader of a group of calls over one vector): only sorts made by
    // that call or after it qualify, so a later group -- the next proof over the same buffer -- never reads an earlier group's sort
    uint64_t msm_seq = 0, msm_leader_seq = 0;
    int msm_slots_in_use = 4;
    char* msm_pinned = nullptr;
    size_t msm_pinned_bytes = 0, msm_pinned_used = 0;   // a ring: `used` is the tail, the oldest pending result's offset the head
    std::vector<czk::MsmPending> msm_pending;
    uint64_t msm_delivered = 0;                         // results delivered so far (absolute index of msm_pending[0])
    std::vector<czk::CtxMark> marks;                    // czk_ctx_mark, oldest first
    uint64_t next_mark = 1;
    std::vector<hipEvent_t> mark_events;                // idle events of retired marks
    // ---- czk_ctx_set_option (core.hip).  The product library knows the first group only; the second group selects kernels that exist in the
    // lab build alone (libczk_hip_lab.so, -DCZK_LAB: the measured-and-rejected variants of EXPERIMENTS.md) and is fixed at its default otherwise.
    bool msm_sort_reuse = false;     // "msm_sort_reuse": CZK_MEM_SAME_SCALARS calls read an earlier call's digit sort when the layouts match (default off: measured, EXPERIMENTS.md section 14)
    bool msm_sort_reuse_any_inf = false;   // lab "msm_sort_reuse_any_inf": ... even when the keys' points at infinity differ (WRONG results: the upper bound of sharing one sort per scalar vector)
    bool msm_sort_onepass = false;   // "msm_sort_onepass": the single-pass digit sort for every call (it is the > 2048-partition fallback anyway)
    bool msm_fixed_c = false;        // "msm_fixed_c": keys registered from now on keep their own window width for short calls (no secondary table sets)
    unsigned msm_c_g1 = 0, msm_c_g2 = 0;   // "msm_window_g1" / "msm_window_g2": primary window width of keys registered from now on (0 = cost model)
    bool msm_launch_split = false;   // (set by msm_enqueue for the launch it is making: a table-free call, lanes = windows)
    int msm_lane_interleave = 0;     // "msm_lane_interleave": lanes per interleave group of the accumulate kernels (msm_acc.h acc_work_item); 0 = the default rule
    int msm_stream_prio = 0;         // "msm_stream_priority": 1 = sort / reduce streams above the accumulate stream, 2 = the reverse (before the first MSM)
    unsigned msm_affine_rounds = 0;  // lab "msm_affine_rounds": R rounds of batched-affine pair additions in front of the G1 bucket accumulation
    bool msm_reduce_sat = false;     // lab "msm_reduce_sat": buckets and their reduction in the saturated form
    bool msm_reduce_sat_g2 = false;  // lab "msm_reduce_sat_g2": the same for G2 only
    int msm_g2_mode = 0;             // lab "msm_g2_mode": 0 = single-lane G2 accumulate kernel (k_accumulate_u2, adopted), 1 = lane pairs <128, 2>, 2 = lane pairs <512, 3> + LDS-limited occupancy
    bool msm_sat = false, msm_sat_g2 = false;   // lab "msm_sat" / "msm_sat_g2": keys registered from now on keep saturated tables and accumulate kernels
    bool msm_no_te = false;          // lab "msm_no_te": G1 keys registered from now on keep the XYZZ kernels (what CZK_MEM_ANY_POINTS does per key)
    long net_create_timeout_ms = 0;  // "net_create_timeout_ms": how long czk_net_create on this context waits for its peers (0 = the communicator's default, 120 s)
    unsigned long long* open_bad = nullptr;   // device counter of czk_fr_spdz_open (allocated once)
    // lab "chaos": schedule perturbation (tests/test_chaos.py).  Non-zero = seed: every stage boundary (the profiling brackets around the sort / accumulate /
    // reduce / NTT / polynomial stages, the result copy) first enqueues a spin kernel of random length on the stage's stream and / or sleeps on the host, and the
    // MSM workspace ring hands out its slots in random order.  Results must not change: every ordering the library relies on has to be an event wait or stream
    // order, never timing.  "chaos_drop_wait" = 1 removes ONE such wait on purpose (the reduce stream's wait for the accumulate kernel:
    // msm.hip), so that the test can show it catches the class of bug it exists for.
    bool ntt_skip_coset_first = false;   // lab "ntt_skip_coset_first": timing experiment (wrong results), see ntt.hip
    unsigned long long chaos = 0;
    int chaos_drop_wait = 0;
    bool ntt_fuse_pairs = false;     // "ntt_fuse_pairs": the witness map's ifft -> coset_fft pairs share a pass where their tiles line up (ntt.hip ifft_coset_fft_device);
                                     // measured + 0.5 % per Groth16 proof (EXPERIMENTS.md section 14), below the 1 % adoption bar: off by default
    bool ntt_gen1 = false;           // "ntt_gen1": first-generation NTT passes (ntt.hip, the small-domain kernels) for every size
    bool profiling = false;
    std::map<std::string, czk::ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t prof_base = nullptr;   // recorded by czk_profile_reset: origin of czk_profile_intervals
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::map<unsigned, czk::DomainTables> domains;
    std::map<unsigned, czk::MixedDomain> mixed_domains;
    czk::DeviceBuf mixed_scratch;   // de-interleaved lanes of the mixed-radix NTT
    czk::DeviceBuf ntt_scratch;   // one lane-batch for the out-of-place NTT passes
    czk::DeviceBuf poly_scratch;  // segment sums of czk_poly_div_linear (poly.hip)
    czk::DeviceBuf share_tab;     // size_inv * w^(-jk) table of czk_fr_gsz_open (share.hip)
    int num_cu = 256;
    size_t lds_per_block = 64 * 1024;   // hipDeviceProp_t::sharedMemPerBlock (gfx950: 160 KiB)
};

// A second (third, ...) set of window tables over a PREFIX of a registered base array, at a narrower window width: short MSMs
// under a long key (KZG commitments of low-degree polynomials under `powers_of_g`, poly-commit/src/kzg10/mod.rs:159-162) would
// otherwise pay the key's 2^(c-1)-bucket reduction per call.  Built on first use by msm.hip (pick_tables), immutable afterwards.
struct czk_table_set {
    unsigned c = 0, W = 0;
    size_t cover = 0;          // points covered = stride between windows
    uint64_t* pts = nullptr;   // W x cover x (12|24) u64
    uint8_t* inf = nullptr;    // W x cover
};

struct czk_bases {
    static constexpr int MAX_EXTRA = 12;
    czk_table_set extra[MAX_EXTRA];
    std::atomic<int> n_extra{0};      // published count: readers scan [0, n_extra) without the lock
    std::atomic<unsigned> nomem_class{0};   // bit c: building a secondary set of width c failed for lack of memory -- not retried per call (pick_tables)
    std::mutex build_mu;              // serialises builders (contexts of several threads may share one handle)
    bool per_call_width = true;       // CZK_MSM_FIXED_C=1 at registration turns the secondary sets off (A/B runs)
    int device = 0;            // GPU ordinal the tables live on (the handle may outlive its context: no ctx pointer is kept)
    int group = 1;
    size_t n = 0;
    unsigned c = 0;            // signed-digit window width chosen at registration
    unsigned W = 0;            // number of windows = ceil(254 / c)
    bool unsat = false;        // window tables hold coordinates * R' (fqu.h), used by k_accumulate_u / k_accumulate_u2
    bool te = false;           // G1 only: window tables hold twisted Edwards niels entries (te.h: 24 u64 per point); buckets and the
                               // reduction run in extended coordinates; `pts_sw0` keeps the original points for secondary table sets
    bool check_wanted = false; // CZK_MEM_CHECK_SUBGROUP: [r] P == infinity is verified at registration; a failing base clears te_wanted
    bool checked = false;      // the check ran at registration; n_bad holds its result
    size_t n_bad = 0;
    bool te_wanted = true;     // false: registered with CZK_MEM_ANY_POINTS (bases need not lie in the prime-order subgroup)
    uint64_t* pts_sw0 = nullptr;   // te: window 0 as registered (n x 12 u64, saturated Montgomery form)
    bool split = false;        // CZK_MEM_NO_TABLES: only window 0 is stored; an MSM runs one bucket set per window (windows become
                               // extra lanes of the same kernels) and the per-window results are combined afterwards
    uint64_t* pts = nullptr;   // device, W x n x (12|24) u64: window w holds 2^(c*w) * P_i, affine Montgomery
    uint8_t* inf = nullptr;    // device, W x n infinity flags (never null)
    // the indices (w n + i) of the primary table's entries at infinity when there are at most INF_LIST_MAX of them (host copy, made at registration): two keys
    // of equal length and layout whose lists are equal drop the same digits, so MSMs over one scalar vector can share a digit sort (CZK_MEM_SAME_SCALARS)
    static constexpr size_t INF_LIST_MAX = 4096;
    bool inf_listed = false;
    std::vector<uint32_t> inf_idx;
};

namespace czk {

int set_err(czk_ctx* ctx, int code, const std::string& msg);

// every buffer argument is host or device memory; the STABLE bit is only meaningful for czk_msm_async
inline bool valid_mem(int mem) { return mem == CZK_MEM_HOST || mem == CZK_MEM_DEVICE; }

// One Fr handed over by HOST pointer (czk.h promises a plain `const uint64_t*`: 8-byte alignment only, e.g. a Rust
// [u64; 4] or a C stack array), so it is read limb by limb -- fp_load's 16-byte vector loads are for device pointers.
inline Fr host_fr(const uint64_t* p) {
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (u32)p[i];
        r.l[2 * i + 1] = (u32)(p[i] >> 32);
    }
    return r;
}

// host <-> device staging for CZK_MEM_HOST callers of the vector entry points
// (device buffers come from a small per-context pool -- stage_take / stage_give -- instead of hipMalloc / hipFree per call:
// a 2^21 x 4-lane NTT from host memory stages 256 MiB, and hipFree alone synchronises the device)
struct Staged {
    czk_ctx* ctx;
    void* dev = nullptr;
    bool owned = false;
    size_t cap = 0;
    int to_device(const void* host, size_t bytes, int mem);
    int to_host(void* host, size_t bytes);
    ~Staged();
};
int ensure_buf(czk_ctx* ctx, DeviceBuf& b, size_t bytes);
// staging pool: a buffer of at least `bytes` (smallest fitting pooled one, else a fresh allocation); give it back when the work
// that uses it has been ENQUEUED on ctx->stream -- every later user enqueues on the same stream, so re-use is ordered
int stage_take(czk_ctx* ctx, size_t bytes, DeviceBuf* out);
void stage_give(czk_ctx* ctx, const DeviceBuf& b);
int get_domain(czk_ctx* ctx, unsigned log_d, DomainTables** out);

// lab build, option "chaos": a random delay on `st` and / or on the calling thread (no-op otherwise, and in the product build)
#ifdef CZK_LAB
void chaos_point(czk_ctx* ctx, hipStream_t st);
unsigned chaos_rand(czk_ctx* ctx);
#else
inline void chaos_point(czk_ctx*, hipStream_t) {}
#endif
// RAII bracket: records an event pair around the launches issued in its scope (no-op unless profiling)
struct ProfScope {
    czk_ctx* ctx;
    const char* name;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(czk_ctx* c, const char* n, hipStream_t s = nullptr);
    ~ProfScope();
};

#define CZK_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return czk::set_err((ctx), CZK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Two device tables that are only ever used together (forward / inverse twiddles): both pointers are published only when both
// allocations succeeded, so a failed second hipMalloc cannot leave a half-initialised pair behind for the next call.
template <class T>
inline int alloc_table_pair(czk_ctx* ctx, T** a, T** b, size_t bytes) {
    void *pa = nullptr, *pb = nullptr;
    if (hipMalloc(&pa, bytes) != hipSuccess) return czk::set_err(ctx, CZK_ERR_NOMEM, "hipMalloc twiddle table");
    if (hipMalloc(&pb, bytes) != hipSuccess) {
        (void)hipFree(pa);
        return czk::set_err(ctx, CZK_ERR_NOMEM, "hipMalloc twiddle table");
    }
    *a = (T*)pa;
    *b = (T*)pb;
    return CZK_OK;
}
// ... and dropped again when the kernels that fill them could not be launched
template <class T>
inline int drop_table_pair(czk_ctx* ctx, T** a, T** b, hipError_t e) {
    (void)hipFree(*a);
    (void)hipFree(*b);
    *a = nullptr;
    *b = nullptr;
    return czk::set_err(ctx, CZK_ERR_HIP, std::string("twiddle table kernels: ") + hipGetErrorString(e));
}

#define CZK_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != CZK_OK) return rc__; \
    } while (0)

// implemented in lanes.hip
void xfer_destroy(czk_ctx* ctx);
int upload_pageable(czk_ctx* ctx, void* dev, const void* host, size_t bytes);
int download_pageable(czk_ctx* ctx, void* host, const void* dev, size_t bytes);
// implemented in ntt.hip
int ntt_device(czk_ctx* ctx, u64* data, unsigned log_d, size_t lanes, int kind, size_t in_len, const u64* src = nullptr, size_t src_stride = 0);
// implemented in ntt_mixed.hip
int get_mixed_domain(czk_ctx* ctx, unsigned k, MixedDomain** out);
int ntt_mixed_device(czk_ctx* ctx, u64* data, unsigned k, size_t lanes, int kind, size_t in_len);
// implemented in msm.hip
int msm_reserve(czk_ctx* ctx, const czk_bases* bases, size_t n_scalars, size_t lanes);   // czk_ctx_reserve (msm.hip)
int ntt_reserve(czk_ctx* ctx, unsigned log_d, size_t lanes);                                   // czk_ctx_reserve (ntt.hip)
int msm_device(czk_ctx* ctx, const czk_bases* bases, const u64* scalars_dev, size_t n_scalars, size_t lanes,
               int scalar_form, u64* out_jac_host, bool blocking, bool scalars_stable, bool same_scalars = false);
int msm_pipeline_init(czk_ctx* ctx);
// Window layout of the signed-digit split (254 bits: 253-bit scalars + the carry).  W(c) = ceil(254 / c) windows of c bits overshoot by
// slack = W c - 254 bits, which the plain layout leaves in the TOP window: with a narrow top window (< 10 bits) every point's top digit lands on
// one of 2^top buckets -- n / 2^top additions in a row on each, the latency of a mid-size MSM (profiles/r04_tiny_msm.txt).  Such widths are
// BALANCED instead: the last `slack` windows are c - 1 bits wide, so every window is full and no bucket sees more than twice the average.
// Widths with a top window of >= 10 bits (every layout chosen for n >= 2^14 points: c = 15, 17, 20, ...) keep the plain layout.
CZK_HD unsigned msm_num_windows(unsigned c) { return (254 + c - 1) / c; }
CZK_HD unsigned msm_full_windows(unsigned c) {   // windows [0, W_hi) are c bits wide, windows [W_hi, W) are c - 1 bits wide
    const unsigned W = msm_num_windows(c), top = 254 - (W - 1) * c, slack = W * c - 254;
    return (top < 10 && slack <= W) ? W - slack : W;
}
CZK_HD unsigned msm_win_bit(unsigned c, unsigned W_hi, unsigned w) { return w * c - (w > W_hi ? w - W_hi : 0); }   // first bit of window w
CZK_HD unsigned msm_win_width(unsigned c, unsigned W_hi, unsigned w) { return w < W_hi ? c : c - 1; }
// sum_w 2^(bit(w)) R[lane][w] on the host (Horner: width(w) doublings per window); src: lanes x W Jacobian triples, out: lanes triples
void host_combine_windows(int group, const char* src, unsigned W, unsigned c, size_t lanes, uint64_t* out);
int msm_pipeline_sync(czk_ctx* ctx);
// lanes per interleave group of the bucket accumulation kernels (msm_acc.h acc_work_item): the option, or the default rule
// Default rule (EXPERIMENTS.md section 14): groups of up to 4 lanes for keys with window tables -- every lane's buckets have the same size
// distribution, so equal ranks are equal work (measured: 4, 6 and 8 lanes per group are equal within noise; the Groth16 step gains 5 %) -- and no
// interleaving on the table-free path, whose lanes are the WINDOWS of a share lane: the narrow top window's buckets are larger, equal ranks are not
// equal work, and interleaving costs 6 - 20 % there (measured).  msm_enqueue notes which kind of call is being launched.
inline unsigned acc_interleave(const czk_ctx* ctx, unsigned lanes) {
    unsigned g = ctx->msm_lane_interleave > 0 ? (unsigned)ctx->msm_lane_interleave : (ctx->msm_launch_split ? 1u : CZK_ACC_INTERLEAVE_DEFAULT);
    if (g > lanes) g = lanes;
    return g ? g : 1;
}
int msm_pinned_take(czk_ctx* ctx, size_t bytes, char** out);   // staging for one host result (a ring over the pinned area; drains when full)
int ctx_mark(czk_ctx* ctx, uint64_t* out);       // czk_ctx_mark / czk_ctx_wait_mark
int ctx_wait_mark(czk_ctx* ctx, uint64_t id);
void msm_pipeline_destroy(czk_ctx* ctx);
int fixed_base_points_device(czk_ctx* ctx, int group, const u64* k_dev, size_t n, u64* out_dev);
void launch_reduce_level_g1(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes);
// (G2: `ub` = buckets and level arrays in u-form, reduced on fq2pu.h's unsaturated lane pairs)
void launch_reduce_level_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                            unsigned lanes, int ub);
void launch_finish_g1(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out);
// tail of the G1 bucket reduction for n_in <= 1024 entries per lane; scratch: lanes * 12 * 512 points, sums: lanes * 12 points
void launch_heavy_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                     u64* buckets, unsigned lanes, const uint8_t* dirty, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap, int unsat, int ubuckets);
// the bucket reduction on u-form buckets (G1: buckets stay in the unsaturated residue system of fqu.h until the last step)
void launch_reduce_level_g1_u(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned L, unsigned scale_dbl, u64* Po, u64* Eo, size_t n_out,
                              unsigned lanes, int te);
void launch_finish_g1_u(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out, int te);
void launch_reduce_tail_g1_u(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out, unsigned lanes,
                             int te);
// G1 in twisted Edwards form (te.h): table conversion at registration, bucket accumulation, over-full buckets
void launch_sw_to_te_niels(hipStream_t st, const u64* aff, const uint8_t* inf, size_t n, u64* scratch, u64* out, u32* bad);
void launch_accumulate_g1_te(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                             size_t sorted_stride, u64* buckets, unsigned lanes);
void launch_heavy_g1_te(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride, u64* buckets,
                        unsigned lanes, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap);
void launch_heavy_g2(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                     u64* buckets, unsigned lanes, const uint8_t* dirty, u32* hdr, u32* items, u32* heavy, u64* partials, u32 cap, int unsat, int ubuckets);
void launch_reduce_tail_g2(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out,
                           unsigned lanes, int ub);
void launch_reduce_tail_g1(hipStream_t st, const u64* P, const u64* E, size_t n_in, unsigned scale_dbl, u64* scratch, u64* sums, u64* out,
                           unsigned lanes);
void launch_finish_g2(hipStream_t st, const u64* P, const u64* E, size_t segs, u64* out, int ub);
// implemented in msm_acc_g1.hip / msm_acc_g2.hip (hot kernels, built with the multiply inlined)
void launch_accumulate_g1(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes);
void launch_accumulate_g1_u_prepare(hipStream_t st, uint8_t* dirty, size_t B, unsigned lanes);
void launch_accumulate_g2_u_prepare(hipStream_t st, uint8_t* dirty, size_t B, unsigned lanes);
void launch_accumulate_g1_u_fixup(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                  u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets);
void launch_accumulate_g2_u_fixup(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                  u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets);
void launch_accumulate_g1_u(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                            size_t sorted_stride, u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets);
void launch_convert_to_u(hipStream_t st, u64* pts, size_t n_coords);
void launch_convert_from_u(hipStream_t st, u64* pts, size_t n_coords);   // the inverse: table coordinates back to the saturated Montgomery form
// batched-affine pre-reduction of the bucket lists (lab/msm_aff.h; G1; lab build only -- the struct stays so that msm_enqueue reads the same either way)
struct AffArgs {
    unsigned rounds = 0, lanes = 0, n_parts = 0, part_shift = 0, part_log = 0;
    size_t B = 0, sorted_stride = 0, S[3] = {0, 0, 0};
    const u32 *sorted = nullptr, *offsets = nullptr, *counts = nullptr;
    void* rec[3] = {nullptr, nullptr, nullptr};       // uint2 records per round
    uint8_t* pend[3] = {nullptr, nullptr, nullptr};   // per-slot "redo with complete formulas" flags
    u32 *off[2] = {nullptr, nullptr}, *cnt[2] = {nullptr, nullptr};   // bucket offsets / counts of level r at index r & 1
    void* lvl[2] = {nullptr, nullptr};                // level arrays, ping-pong
    void* scratch = nullptr;                          // running products of the resident waves
};
void aff_plan(size_t total0, size_t B, unsigned rounds, size_t* S);
size_t aff_scratch_bytes(czk_ctx* ctx);
void launch_affine_build_g1(hipStream_t st, const AffArgs& a);
void launch_affine_accumulate_g1(czk_ctx* ctx, hipStream_t st, const AffArgs& a, const u64* pts, const u32* perm, u64* buckets, uint8_t* dirty);
void launch_accumulate_g1_u_fixup_lvl(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, size_t B, size_t sorted_stride,
                                      u64* buckets, unsigned lanes, uint8_t* dirty, const void* lvl);
void launch_accumulate_g2_u(czk_ctx* ctx, hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                            size_t sorted_stride, u64* buckets, unsigned lanes, uint8_t* dirty, int ubuckets);
void launch_accumulate_g2(hipStream_t st, const u64* pts, const u32* sorted, const u32* offsets, const u32* counts, const u32* perm, size_t B,
                          size_t sorted_stride, u64* buckets, unsigned lanes);

}  // namespace czk

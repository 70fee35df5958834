// czk.hpp -- C++ host layer over the C ABI (czk.h), mirroring the reference's Rust surface for this path so
// that calling code reads like the reference's.  Header-only; link with libczk_hip.so.
//
//   reference (Rust)                                                         here (C++)
//   ------------------------------------------------------------------------------------------------------------
//   ark_poly::EvaluationDomain::new(n) -> Option<Self>                       Radix2EvaluationDomain::create(n) -> std::optional
//     (algebra/poly/src/domain/radix2/mod.rs:51-82)
//   domain.{fft,ifft,coset_fft,coset_ifft}_in_place(&mut Vec<T>)             same names, T = Fr or MpcField (share lanes)
//     (algebra/poly/src/domain/mod.rs:79,90,139,155)
//   domain.divide_by_vanishing_poly_on_coset_in_place(&mut [F])              same name (domain/mod.rs:184-191)
//   VariableBaseMSM::multi_scalar_mul(&[G], &[BigInt]) -> G::Projective      VariableBaseMSM::multi_scalar_mul(bases, scalars)
//     (algebra/ec/src/msm/variable_base.rs:12-15)
//   AffineCurve::multi_scalar_mul(&[Self], &[Fr]) -> Projective              G1Affine::multi_scalar_mul / G2Affine::multi_scalar_mul
//     (algebra/ec/src/lib.rs:300-311)
//   MpcField::{Public, Shared}, SpdzFieldShare{sh, mac}                      MpcField{shared, sh, mac}
//     (mpc-algebra/src/wire/field.rs:27-30, share/spdz.rs:50-53)
//   GroupShare::multi_scale_pub_group(bases, &[share]) (SPDZ)                SpdzGroupShare::multi_scale_pub_group
//     (mpc-algebra/src/share/spdz.rs:440-446)
//   R1CStoQAP::witness_map (mpc-snarks/src/groth/r1cs_to_qap.rs:47-113)      R1CStoQAP::witness_map(domain, a, b, c [, batch_product])
//   MpcMultiNet::{party_id, n_parties, am_king, broadcast, send_to_king,      Net (czk_net: RCCL between GPUs, or shared memory between
//     recv_from_king, stats} (mpc-net/src/multi.rs:145-242, lib.rs:30-76)      processes of one node), on DeviceLanes and on host bytes
//   SpdzFieldShare / AdditiveFieldShare / GszFieldShare::batch_open           Net::spdz_batch_open / add_batch_open / gsz_batch_open
//     (share/spdz.rs:166-185, add.rs:256-259, gsz20/mod.rs:286-300)
//   Vec<T> that stays alive across the witness map and feeds the h MSM        DeviceLanes (czk_lanes: the vector lives in HBM; every
//     (r1cs_to_qap.rs:66-110, groth/prover.rs:104)                            domain / MSM / pointwise call below has a DeviceLanes form)
//
// Error behaviour: where the reference returns None the mirror returns std::nullopt; where it `assert!`s /
// `unwrap()`s the mirror throws czk::Panic carrying czk_last_error() (a Rust shim would `expect()` the status).
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "czk.h"

namespace czk {

struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// Plain-old-data views of the reference's values (little-endian u64 limbs, Montgomery form).
struct Fr { uint64_t l[4]; };
struct BigInteger256 { uint64_t l[4]; };
struct Fq { uint64_t l[6]; };
struct Fq2 { Fq c0, c1; };
struct G1Projective { Fq x, y, z; };
struct G2Projective { Fq2 x, y, z; };

class Context {
  public:
    explicit Context(int device = 0, void* hip_stream = nullptr) {
        int rc = czk_ctx_create(&ctx_, device, hip_stream);
        if (rc != CZK_OK) throw Panic(rc, "czk_ctx_create failed (no GPU visible? the product path has no CPU fallback)");
    }
    ~Context() { czk_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    czk_ctx* raw() const { return ctx_; }
    void check(int rc) const {
        if (rc != CZK_OK) throw Panic(rc, czk_last_error(ctx_));
    }
    void sync() const { check(czk_ctx_sync(ctx_)); }
    // czk_ctx_mark / czk_ctx_wait_mark: wait for the work enqueued up to a point (and take delivery of its MSM results / deferred downloads) while
    // later calls keep running -- the transcript points of the polynomial provers (tools/polyvm_host.hpp)
    uint64_t mark() const {
        uint64_t m = 0;
        check(czk_ctx_mark(ctx_, &m));
        return m;
    }
    void wait_mark(uint64_t m) const { check(czk_ctx_wait_mark(ctx_, m)); }
    // czk_ctx_reserve: twiddle tables of a 2^ntt_log_d domain (0: none) and the MSM workspaces for calls of `n_scalars` x `msm_lanes` on
    // `bases` (nullptr: none), at key load instead of inside the first proof
    void reserve(unsigned ntt_log_d, size_t ntt_lanes, const czk_bases* bases = nullptr, size_t n_scalars = 0, size_t msm_lanes = 0) const {
        check(czk_ctx_reserve(ctx_, ntt_log_d, ntt_lanes, bases, n_scalars, msm_lanes));
    }

  private:
    czk_ctx* ctx_ = nullptr;
};

// `lanes` share-vector components of `capacity` Fr each, resident on the context's GPU (czk_lanes).  `len` plays the role of
// Vec::len(): elements [len, capacity) are the zero padding `resize(size, T::zero())` would append (radix2/mod.rs:100-101) and
// are never read -- the transforms zero-extend.  Uploads come from pageable host memory (a Rust Vec); nothing else moves.
class DeviceLanes {
  public:
    DeviceLanes(const Context& ctx, size_t lanes, size_t capacity) : ctx_(&ctx), len(capacity) {
        ctx.check(czk_lanes_alloc(ctx.raw(), lanes, capacity, &h_));
    }
    ~DeviceLanes() { czk_lanes_free(h_); }
    DeviceLanes(DeviceLanes&& o) noexcept : ctx_(o.ctx_), h_(o.h_), len(o.len) { o.h_ = nullptr; }
    DeviceLanes(const DeviceLanes&) = delete;
    DeviceLanes& operator=(const DeviceLanes&) = delete;
    size_t lanes() const { return czk_lanes_count(h_); }
    size_t capacity() const { return czk_lanes_len(h_); }
    uint64_t* data(size_t lane = 0, size_t elem = 0) const { return czk_lanes_data(h_, lane, elem); }
    czk_lanes* raw() const { return h_; }
    const Context& ctx() const { return *ctx_; }
    void upload(size_t lane, size_t elem, const Fr* host, size_t n) { ctx_->check(czk_lanes_upload(ctx_->raw(), h_, lane, elem, n ? host->l : nullptr, n)); }
    void upload(size_t lane, const std::vector<Fr>& v) { upload(lane, 0, v.data(), v.size()); }
    void download(size_t lane, size_t elem, Fr* host, size_t n) const { ctx_->check(czk_lanes_download(ctx_->raw(), h_, lane, elem, n ? host->l : nullptr, n)); }
    // czk_lanes_download_deferred: `host` is filled when a later mark is waited for / at the next sync
    void download_deferred(size_t lane, size_t elem, Fr* host, size_t n) const {
        ctx_->check(czk_lanes_download_deferred(ctx_->raw(), h_, lane, elem, n ? host->l : nullptr, n));
    }
    std::vector<Fr> to_host(size_t lane) const {
        std::vector<Fr> v(len);
        download(lane, 0, v.data(), len);
        return v;
    }
    void copy_from(size_t lane, size_t elem, const DeviceLanes& src, size_t src_lane, size_t src_elem, size_t n) {
        ctx_->check(czk_lanes_copy(ctx_->raw(), h_, lane, elem, src.h_, src_lane, src_elem, n));
    }
    DeviceLanes clone() const {   // `let mut ab = a.clone()`
        DeviceLanes c(*ctx_, lanes(), capacity());
        c.len = len;
        c.copy_from(0, 0, *this, 0, 0, lanes() * capacity());
        return c;
    }

  private:
    const Context* ctx_;
    czk_lanes* h_ = nullptr;

  public:
    size_t len;   // logical Vec length of every lane (<= capacity)
};

// mpc-net's MpcMultiNet for parties that are czk contexts (include/czk.h "mpc-net between GPUs"): the exchanges of an open run on the
// context's stream, on lanes that stay in HBM.  One Net per (Context, party); destroy it before its Context.
//   transport CZK_NET_RCCL: one party per GPU, id = Net::unique_id(CZK_NET_RCCL) of rank 0, handed to the other parties by the launcher
//   transport CZK_NET_SHM : parties are processes of one node (several may share a GPU), id = any 1..32 bytes the launcher chose
class Net {
  public:
    struct Stats { uint64_t bytes_sent, bytes_recv, broadcasts, to_king, from_king; };   // mpc-net/src/lib.rs Stats
    Net(const Context& ctx, int transport, int rank, int world, const std::vector<uint8_t>& id) : ctx_(&ctx) {
        ctx.check(czk_net_create(ctx.raw(), transport, rank, world, id.data(), id.size(), &n_));
    }
    ~Net() { czk_net_destroy(n_); }
    Net(const Net&) = delete;
    Net& operator=(const Net&) = delete;
    static std::vector<uint8_t> unique_id(int transport) {
        std::vector<uint8_t> id(CZK_NET_UNIQUE_ID_BYTES);
        size_t len = 0;
        int rc = czk_net_unique_id(transport, id.data(), id.size(), &len);
        if (rc != CZK_OK) throw Panic(rc, "czk_net_unique_id failed (RCCL transport: librccl.so.1 could not be loaded)");
        id.resize(len);
        return id;
    }
    czk_net* raw() const { return n_; }
    const Context& ctx() const { return *ctx_; }
    void check(int rc) const {
        if (rc != CZK_OK) throw Panic(rc, czk_net_last_error(n_));
    }
    size_t party_id() const { return (size_t)czk_net_rank(n_); }     // MpcNet::party_id
    size_t n_parties() const { return (size_t)czk_net_world(n_); }   // MpcNet::n_parties
    bool am_king() const { return party_id() == 0; }                  // MpcNet::am_king
    void set_option(const char* name, long value) const { check(czk_net_set_option(n_, name, value)); }
    Stats stats() const {
        uint64_t s[5];
        check(czk_net_stats(n_, s));
        return Stats{s[0], s[1], s[2], s[3], s[4]};
    }
    void reset_stats() const { czk_net_stats_reset(n_); }
    void barrier() const { check(czk_net_barrier(n_)); }
    // MpcNet::broadcast_bytes / send_bytes_to_king / recv_bytes_from_king on host bytes (mpc-net/src/lib.rs:44-61)
    std::vector<std::vector<uint8_t>> broadcast_bytes(const std::vector<uint8_t>& out) const {
        std::vector<uint8_t> flat(out.size() * n_parties());
        check(czk_net_broadcast(n_, out.data(), out.size(), flat.data(), CZK_MEM_HOST));
        return split(flat, out.size());
    }
    std::optional<std::vector<std::vector<uint8_t>>> send_bytes_to_king(const std::vector<uint8_t>& out) const {
        std::vector<uint8_t> flat(am_king() ? out.size() * n_parties() : 0);
        check(czk_net_send_to_king(n_, out.data(), out.size(), am_king() ? flat.data() : nullptr, CZK_MEM_HOST));
        if (!am_king()) return std::nullopt;
        return split(flat, out.size());
    }
    std::vector<uint8_t> recv_bytes_from_king(const std::optional<std::vector<std::vector<uint8_t>>>& out, size_t m) const {
        std::vector<uint8_t> flat, mine(m);
        if (am_king()) {
            if (!out || out->size() != n_parties()) throw Panic(CZK_ERR_ARG, "recv_bytes_from_king: the king passes one buffer per party");
            for (const auto& b : *out) {
                if (b.size() != m) throw Panic(CZK_ERR_ARG, "assertion failed: bytes_out[id].len() == m");   // multi.rs:224
                flat.insert(flat.end(), b.begin(), b.end());
            }
        }
        check(czk_net_recv_from_king(n_, am_king() ? flat.data() : nullptr, m, mine.data(), CZK_MEM_HOST));
        return mine;
    }
    // The batch opens on lanes that live on the GPU.  Each opens `n` elements starting at element 0 of the given lanes and writes the
    // opened (public) vector to `out`; a failed check panics like the reference's assert.
    //   SpdzFieldShare::batch_open (share/spdz.rs:166-185): `shares` lane `sh_lane` = sh, lane `sh_lane + 1` = mac.  `commit` (the default, as in the
    //   reference: spdz.rs:179 sends dx_ts through Net::atomic_broadcast, channel.rs:50-75) runs the commit-then-open round; false is an explicit
    //   opt-out for measurements of the two-round form and must be reported as such
    void spdz_batch_open(const DeviceLanes& shares, size_t sh_lane, const Fr& mac_share, size_t n, uint64_t* out, bool commit = true) const {
        uint64_t bad = 0;
        check(czk_spdz_batch_open(n_, shares.data(sh_lane), shares.data(sh_lane + 1), mac_share.l, n, out, commit ? CZK_OPEN_COMMIT : 0, &bad));
        if (bad) throw Panic(CZK_ERR_CHECK, "assertion failed: sum.is_zero() (SPDZ MAC check, share/spdz.rs:183) on " + std::to_string(bad) + " values");
    }
    //   AdditiveFieldShare::batch_open (share/add.rs:256-259)
    void add_batch_open(const uint64_t* val, size_t n, uint64_t* out) const { check(czk_add_batch_open(n_, val, n, out)); }
    //   GszFieldShare::batch_open (share/gsz20/mod.rs:286-300) with one degree bound for the whole vector
    void gsz_batch_open(const uint64_t* val, size_t n, unsigned degree, uint64_t* out) const {
        uint64_t bad = 0;
        check(czk_gsz_batch_open(n_, val, n, nullptr, degree, out, &bad));
        if (bad) throw Panic(CZK_ERR_CHECK, "assertion failed: p.degree() <= d (share/gsz20/mod.rs:452) on " + std::to_string(bad) + " values");
    }
    //   gsz20::batch_king_compute(shares, new_degree, |r| r) (share/gsz20/mod.rs:494-527): the degree reduction inside batch_mult
    void gsz_batch_king_compute(const uint64_t* val, size_t n, unsigned degree, uint64_t* out) const {
        uint64_t bad = 0;
        check(czk_gsz_batch_king_compute(n_, val, n, nullptr, degree, out, &bad));
        if (bad) throw Panic(CZK_ERR_CHECK, "assertion failed: p.degree() <= d (king, share/gsz20/mod.rs:452) on " + std::to_string(bad) + " values");
    }

  private:
    static std::vector<std::vector<uint8_t>> split(const std::vector<uint8_t>& flat, size_t m) {
        std::vector<std::vector<uint8_t>> r;
        for (size_t p = 0; p * m < flat.size() || (m == 0 && p == 0); p++) {
            r.emplace_back(flat.begin() + p * m, flat.begin() + (p + 1) * m);
            if (m == 0) break;
        }
        return r;
    }
    const Context* ctx_;
    czk_net* n_ = nullptr;
};

// mpc-algebra/src/wire/field.rs:27-30 -- MpcField<Fr, SpdzFieldShare<Fr>>: Public(x) or Shared{sh, mac}
struct MpcField {
    bool shared = false;
    Fr sh{};   // Public: the value; Shared: the additive share
    Fr mac{};  // Shared only: the MAC share (share/spdz.rs:50-53)
};

// SoA lanes of a share vector.  Public(x) entries are lifted to (king ? x : 0) on BOTH lanes, which is what the
// reference's `shift` does the first time a public value meets a share (share/spdz.rs:204-208, add.rs:141-146);
// all butterfly operations are lane-wise linear, so the lifted lanes give the same sh / mac vectors.
struct ShareLanes {
    std::vector<Fr> data;   // 2 x D: lane 0 = sh, lane 1 = mac
    size_t len = 0;
};
inline ShareLanes unpack_lanes(const std::vector<MpcField>& v, size_t domain_size, bool am_king) {
    ShareLanes out;
    out.len = v.size();
    out.data.assign(2 * domain_size, Fr{});
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i].shared) {
            out.data[i] = v[i].sh;
            out.data[domain_size + i] = v[i].mac;
        } else if (am_king) {
            out.data[i] = v[i].sh;
            out.data[domain_size + i] = v[i].sh;
        }
    }
    return out;
}
inline void repack_lanes(const ShareLanes& lanes, size_t domain_size, std::vector<MpcField>& v) {
    v.resize(domain_size);
    for (size_t i = 0; i < domain_size; i++) {
        v[i].shared = true;   // Reveal::from_add_shared (mpc-algebra/src/reveal.rs:15-43)
        v[i].sh = lanes.data[i];
        v[i].mac = lanes.data[domain_size + i];
    }
}

// algebra/poly/src/domain/radix2/mod.rs:23-117
class Radix2EvaluationDomain {
  public:
    // EvaluationDomain::new: None when log2(size) > TWO_ADICITY (radix2/mod.rs:61-63)
    static std::optional<Radix2EvaluationDomain> create(const Context& ctx, size_t num_coeffs, bool am_king = true) {
        size_t size = 1;
        unsigned lg = 0;
        while (size < num_coeffs) {
            size <<= 1;
            lg++;
        }
        Radix2EvaluationDomain d(ctx, am_king);
        d.size_ = size;
        d.log_size_of_group = lg;
        uint64_t k[24];
        if (czk_domain_constants(ctx.raw(), lg, k) != CZK_OK) return std::nullopt;
        auto cp = [&](Fr& f, int idx) { for (int i = 0; i < 4; i++) f.l[i] = k[4 * idx + i]; };
        cp(d.size_inv, 0); cp(d.group_gen, 1); cp(d.group_gen_inv, 2); cp(d.generator, 3); cp(d.generator_inv, 4); cp(d.vanishing_inv_, 5);
        return d;
    }
    size_t size() const { return size_; }

    void fft_in_place(std::vector<Fr>& coeffs) const { run(coeffs, CZK_FFT); }
    void ifft_in_place(std::vector<Fr>& evals) const { run(evals, CZK_IFFT); }
    void coset_fft_in_place(std::vector<Fr>& coeffs) const { run(coeffs, CZK_COSET_FFT); }
    void coset_ifft_in_place(std::vector<Fr>& evals) const { run(evals, CZK_COSET_IFFT); }
    // T = MpcField: two Fr lanes per vector
    void fft_in_place(std::vector<MpcField>& v) const { run_shared(v, CZK_FFT); }
    void ifft_in_place(std::vector<MpcField>& v) const { run_shared(v, CZK_IFFT); }
    void coset_fft_in_place(std::vector<MpcField>& v) const { run_shared(v, CZK_COSET_FFT); }
    void coset_ifft_in_place(std::vector<MpcField>& v) const { run_shared(v, CZK_COSET_IFFT); }

    // T = share lanes resident on the GPU: every lane of `v` is transformed in place, nothing crosses PCIe
    void fft_in_place(DeviceLanes& v) const { run_device(v, CZK_FFT); }
    void ifft_in_place(DeviceLanes& v) const { run_device(v, CZK_IFFT); }
    void coset_fft_in_place(DeviceLanes& v) const { run_device(v, CZK_COSET_FFT); }
    void coset_ifft_in_place(DeviceLanes& v) const { run_device(v, CZK_COSET_IFFT); }
    // the non-in-place forms (domain/mod.rs:72-76, 83-87, 130-134, 146-150: `coeffs.to_vec()` + the in-place transform): the copy is the first
    // pass's load (czk_ntt_fr_to); `v` is left as it was and may be shorter or longer-strided than the domain
    DeviceLanes fft(const DeviceLanes& v) const { return run_device_to(v, CZK_FFT); }
    DeviceLanes ifft(const DeviceLanes& v) const { return run_device_to(v, CZK_IFFT); }
    DeviceLanes coset_fft(const DeviceLanes& v) const { return run_device_to(v, CZK_COSET_FFT); }
    DeviceLanes coset_ifft(const DeviceLanes& v) const { return run_device_to(v, CZK_COSET_IFFT); }
    void divide_by_vanishing_poly_on_coset_in_place(DeviceLanes& evals) const {
        const size_t n = evals.lanes() * evals.capacity();
        // czk_fr_vec_scale reads its constant from the vector's memory space: the device copy of Z(g)^-1 is made ONCE per domain and kept
        // (a per-call allocation would cost a hipMalloc / hipFree -- a device-wide synchronisation -- on the path that never leaves HBM)
        if (!vanishing_inv_dev_) {
            vanishing_inv_dev_ = std::make_shared<DeviceLanes>(*ctx_, 1, 1);
            vanishing_inv_dev_->upload(0, 0, &vanishing_inv_, 1);
        }
        ctx_->check(czk_fr_vec_scale(ctx_->raw(), evals.data(), vanishing_inv_dev_->data(), evals.data(), n, CZK_MEM_DEVICE));
    }
    const Context& ctx() const { return *ctx_; }

    // domain/mod.rs:184-191
    void divide_by_vanishing_poly_on_coset_in_place(std::vector<Fr>& evals) const {
        ctx_->check(czk_fr_vec_scale(ctx_->raw(), evals[0].l, vanishing_inv_.l, evals[0].l, evals.size(), CZK_MEM_HOST));
    }

    unsigned log_size_of_group = 0;
    Fr size_inv{}, group_gen{}, group_gen_inv{}, generator{}, generator_inv{};

  private:
    Radix2EvaluationDomain(const Context& c, bool king) : ctx_(&c), am_king_(king) {}
    void run(std::vector<Fr>& v, int kind) const {
        if (v.size() > size_) throw Panic(CZK_ERR_SIZE, "assertion failed: coeffs.len() <= self.size()");   // radix2/mod.rs:100
        size_t in_len = v.size();
        v.resize(size_, Fr{});                                                                              // :101
        ctx_->check(czk_ntt_fr(ctx_->raw(), v[0].l, log_size_of_group, 1, kind, in_len, CZK_MEM_HOST));
    }
    void run_device(DeviceLanes& v, int kind) const {
        if (v.len > size_ || v.capacity() != size_) throw Panic(CZK_ERR_SIZE, "assertion failed: coeffs.len() <= self.size()");   // radix2/mod.rs:100
        ctx_->check(czk_ntt_fr(ctx_->raw(), v.data(), log_size_of_group, v.lanes(), kind, v.len, CZK_MEM_DEVICE));
        v.len = size_;                                                                                                              // :101
    }
    DeviceLanes run_device_to(const DeviceLanes& v, int kind) const {
        if (v.len > size_) throw Panic(CZK_ERR_SIZE, "assertion failed: coeffs.len() <= self.size()");   // radix2/mod.rs:100
        DeviceLanes out(*ctx_, v.lanes(), size_);
        ctx_->check(czk_ntt_fr_to(ctx_->raw(), v.data(), v.capacity(), out.data(), log_size_of_group, v.lanes(), kind, v.len, CZK_MEM_DEVICE));
        return out;
    }
    void run_shared(std::vector<MpcField>& v, int kind) const {
        if (v.size() > size_) throw Panic(CZK_ERR_SIZE, "assertion failed: coeffs.len() <= self.size()");
        ShareLanes lanes = unpack_lanes(v, size_, am_king_);
        ctx_->check(czk_ntt_fr(ctx_->raw(), lanes.data[0].l, log_size_of_group, 2, kind, lanes.len, CZK_MEM_HOST));
        repack_lanes(lanes, size_, v);
    }
    const Context* ctx_;
    bool am_king_;
    size_t size_ = 0;
    Fr vanishing_inv_{};
    mutable std::shared_ptr<DeviceLanes> vanishing_inv_dev_;   // device copy of vanishing_inv_, made on first use
};

// A proving-key query pinned on the GPU (groth16/src/data_structures.rs:132-149).
template <int GROUP>
class Bases {
  public:
    // mem_flags: 0, CZK_MEM_NO_TABLES for a key that is used once, CZK_MEM_ANY_POINTS for points of unknown origin (complete XYZZ formulas),
    // CZK_MEM_CHECK_SUBGROUP to verify [r] P = 0 at registration and fall back to them when a base fails (see czk.h)
    Bases(const Context& ctx, const uint64_t* xy, const uint8_t* inf, size_t n, int mem_flags = 0) : ctx_(&ctx) {
        ctx.check(czk_bases_register(ctx.raw(), GROUP, xy, inf, n, CZK_MEM_HOST | mem_flags, &b_));
    }
    ~Bases() { czk_bases_release(b_); }
    Bases(const Bases&) = delete;
    Bases& operator=(const Bases&) = delete;
    size_t len() const { return czk_bases_len(b_); }
    // GroupAffine::is_in_correct_subgroup_assuming_on_curve over the whole query (short_weierstrass_jacobian.rs:131): number of bases that fail
    size_t bases_outside_subgroup() const {
        size_t bad = 0;
        ctx_->check(czk_bases_check_subgroup(ctx_->raw(), b_, &bad));
        return bad;
    }
    czk_bases* raw() const { return b_; }
    const Context& ctx() const { return *ctx_; }

  private:
    const Context* ctx_;
    czk_bases* b_ = nullptr;
};
using G1Bases = Bases<CZK_G1>;
using G2Bases = Bases<CZK_G2>;

// algebra/ec/src/msm/variable_base.rs:12-106
struct VariableBaseMSM {
    static G1Projective multi_scalar_mul(const G1Bases& bases, const std::vector<BigInteger256>& scalars) {
        G1Projective out;
        bases.ctx().check(czk_msm(bases.ctx().raw(), bases.raw(), scalars.empty() ? nullptr : scalars[0].l, scalars.size(), 1,
                                  CZK_SCALAR_CANONICAL, CZK_MEM_HOST, out.x.l));
        return out;
    }
    static G2Projective multi_scalar_mul(const G2Bases& bases, const std::vector<BigInteger256>& scalars) {
        G2Projective out;
        bases.ctx().check(czk_msm(bases.ctx().raw(), bases.raw(), scalars.empty() ? nullptr : scalars[0].l, scalars.size(), 1,
                                  CZK_SCALAR_CANONICAL, CZK_MEM_HOST, out.x.c0.l));
        return out;
    }
};

// algebra/ec/src/lib.rs:300-311 -- AffineCurve::multi_scalar_mul: Fr scalars, into_repr on the way in
struct G1Affine {
    static G1Projective multi_scalar_mul(const G1Bases& bases, const std::vector<Fr>& scalars) {
        G1Projective out;
        bases.ctx().check(czk_msm(bases.ctx().raw(), bases.raw(), scalars.empty() ? nullptr : scalars[0].l, scalars.size(), 1,
                                  CZK_SCALAR_MONTGOMERY, CZK_MEM_HOST, out.x.l));
        return out;
    }
};
struct G2Affine {
    static G2Projective multi_scalar_mul(const G2Bases& bases, const std::vector<Fr>& scalars) {
        G2Projective out;
        bases.ctx().check(czk_msm(bases.ctx().raw(), bases.raw(), scalars.empty() ? nullptr : scalars[0].l, scalars.size(), 1,
                                  CZK_SCALAR_MONTGOMERY, CZK_MEM_HOST, out.x.c0.l));
        return out;
    }
};

// The same call on share lanes that are already on the GPU (the `h` vector out of the witness map, prover.rs:104; the witness /
// assignment lanes, :108-156): `lanes` scalar vectors of `n_scalars` Fr starting at scalars.data(lane0), one result per lane.
// Enqueue-only (czk_msm_async): consecutive MSMs pipeline; `out` is valid after ctx.sync().  stable = the caller will not
// overwrite the scalars before that sync (CZK_MEM_STABLE); same_scalars (with stable) = they are the scalars of the previous call, whose digit
// sort the library may take over (CZK_MEM_SAME_SCALARS: create_proof's a, b_g1 and b_g2 all take `assignment`, prover.rs:130-166).
template <int GROUP, class Projective>
inline void multi_scalar_mul_async(const Bases<GROUP>& bases, const DeviceLanes& scalars, size_t n_scalars, Projective* out, bool stable = false,
                                   bool same_scalars = false) {
    static_assert(sizeof(Projective) == (GROUP == CZK_G1 ? 18 : 36) * 8, "Projective does not match the group");
    bases.ctx().check(czk_msm_async(bases.ctx().raw(), bases.raw(), scalars.data(), n_scalars, scalars.lanes(), CZK_SCALAR_MONTGOMERY,
                                    CZK_MEM_DEVICE | (stable ? CZK_MEM_STABLE : 0) | (same_scalars ? CZK_MEM_SAME_SCALARS : 0), reinterpret_cast<uint64_t*>(out)));
}

// mpc-algebra/src/share/spdz.rs:440-446 -- SPDZ multi_scale_pub_group.  The reference builds BOTH scalar vectors from `s.sh.val`
// (:441 and :442), so its second MSM repeats the first bit for bit: one MSM is run and its result returned for `sh` and `mac` --
// identical to the reference's output at half its work.  (A caller with distinct MAC scalars uses multi_scale_pub_group_lanes.)
struct SpdzGroupShareG1 {
    G1Projective sh, mac;
    static SpdzGroupShareG1 multi_scale_pub_group(const G1Bases& bases, const std::vector<MpcField>& scalars) {
        std::vector<Fr> shares(scalars.size());
        for (size_t i = 0; i < scalars.size(); i++) shares[i] = scalars[i].sh;
        G1Projective r = G1Affine::multi_scalar_mul(bases, shares);
        return SpdzGroupShareG1{r, r};
    }
    // two scalar vectors over the same bases in ONE launch (the bases are gathered once): sh from .sh, mac from .mac
    static SpdzGroupShareG1 multi_scale_pub_group_lanes(const G1Bases& bases, const std::vector<MpcField>& scalars) {
        std::vector<Fr> lanes(2 * scalars.size());
        for (size_t i = 0; i < scalars.size(); i++) {
            lanes[i] = scalars[i].sh;
            lanes[scalars.size() + i] = scalars[i].mac;
        }
        G1Projective out[2];
        bases.ctx().check(czk_msm(bases.ctx().raw(), bases.raw(), lanes.empty() ? nullptr : lanes[0].l, scalars.size(), 2,
                                  CZK_SCALAR_MONTGOMERY, CZK_MEM_HOST, out[0].x.l));
        return SpdzGroupShareG1{out[0], out[1]};
    }
};

// poly-commit/src/kzg10/mod.rs:141-193 -- KZG10::commit for a single prover: MSM over powers_of_g plus, when hiding, an
// MSM over powers_of_gamma_g joined with add_assign_mixed (Marlin / Plonk reach the MSM kernel through this).
struct KZG10 {
    static G1Projective commit(const G1Bases& powers_of_g, const std::vector<Fr>& coeffs, const G1Bases* powers_of_gamma_g = nullptr,
                               const std::vector<Fr>* blinding_coeffs = nullptr) {
        if (coeffs.size() > powers_of_g.len()) throw Panic(CZK_ERR_SIZE, "TooManyCoefficients (check_degree_is_too_large)");
        G1Projective commitment = G1Affine::multi_scalar_mul(powers_of_g, coeffs);
        if (powers_of_gamma_g && blinding_coeffs) {
            G1Projective random_commitment = G1Affine::multi_scalar_mul(*powers_of_gamma_g, *blinding_coeffs);
            uint64_t aff[12];
            uint8_t inf = 0;
            const Context& ctx = powers_of_g.ctx();
            ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, random_commitment.x.l, 1, aff, &inf));     // .into_affine()
            ctx.check(czk_jac_add_mixed(ctx.raw(), CZK_G1, commitment.x.l, aff, inf, commitment.x.l));  // add_assign_mixed
        }
        return commitment;
    }

    // KZG10::compute_witness_polynomial (kzg10/mod.rs:200-224): p / (X - point); the remainder p(point) is dropped
    // there and returned here because open() also needs Polynomial::evaluate (:247).
    static std::vector<Fr> compute_witness_polynomial(const Context& ctx, const std::vector<Fr>& p, const Fr& point, Fr* evaluation = nullptr) {
        std::vector<Fr> w(p.size() > 1 ? p.size() - 1 : 0);
        Fr rem{{0, 0, 0, 0}};
        ctx.check(czk_poly_div_linear(ctx.raw(), p.empty() ? nullptr : p[0].l, p.size(), 1, point.l, w.empty() ? nullptr : w[0].l, rem.l, CZK_MEM_HOST));
        if (evaluation) *evaluation = rem;
        return w;
    }
    // Polynomial::evaluate (algebra/poly/src/polynomial/univariate/dense.rs:59-96; kzg10/mod.rs:246, :525): p(point), no quotient written
    static Fr evaluate(const Context& ctx, const std::vector<Fr>& p, const Fr& point) {
        Fr v{{0, 0, 0, 0}};
        ctx.check(czk_poly_evaluate(ctx.raw(), p.empty() ? nullptr : p[0].l, p.size(), 1, point.l, v.l, CZK_MEM_HOST));
        return v;
    }
    // KZG10::open_with_witness_polynomial (kzg10/mod.rs:225-265): proof.w before into_affine; with hiding the second MSM
    // runs over powers_of_gamma_g and random_v = blinding_p(point).
    struct Proof {
        G1Projective w;
        std::optional<Fr> random_v;
    };
    static Proof open(const G1Bases& powers_of_g, const std::vector<Fr>& p, const Fr& point, const G1Bases* powers_of_gamma_g = nullptr,
                      const std::vector<Fr>* blinding_p = nullptr) {
        const Context& ctx = powers_of_g.ctx();
        std::vector<Fr> witness = compute_witness_polynomial(ctx, p, point);
        if (witness.size() > powers_of_g.len()) throw Panic(CZK_ERR_SIZE, "TooManyCoefficients (check_degree_is_too_large)");
        Proof proof{G1Affine::multi_scalar_mul(powers_of_g, witness), std::nullopt};
        if (powers_of_gamma_g && blinding_p) {
            Fr v{{0, 0, 0, 0}};
            std::vector<Fr> hiding = compute_witness_polynomial(ctx, *blinding_p, point, &v);
            G1Projective hw = G1Affine::multi_scalar_mul(*powers_of_gamma_g, hiding);
            ctx.check(czk_jac_add(ctx.raw(), CZK_G1, proof.w.x.l, hw.x.l, proof.w.x.l));       // w += ...
            proof.random_v = v;
        }
        return proof;
    }
};

// One of ConstraintMatrices::{a, b, c} (Vec<Vec<(F, usize)>>, ark-relations) pinned on the GPU in CSR form, and
// evaluate_constraint over all of its rows (mpc-snarks/src/groth/r1cs_to_qap.rs:12-42, :70-77, :95-100).
struct ConstraintMatrix {
    const Context* ctx_;
    czk_r1cs_matrix* h_ = nullptr;
    size_t rows_ = 0;
    ConstraintMatrix(const Context& ctx, const std::vector<std::vector<std::pair<Fr, size_t>>>& rows, size_t n_vars) : ctx_(&ctx), rows_(rows.size()) {
        std::vector<uint64_t> row_ptr(rows.size() + 1, 0);
        std::vector<uint32_t> col;
        std::vector<Fr> coeff;
        for (size_t i = 0; i < rows.size(); i++) {
            for (const auto& term : rows[i]) {
                coeff.push_back(term.first);
                col.push_back((uint32_t)term.second);
            }
            row_ptr[i + 1] = col.size();
        }
        ctx.check(czk_r1cs_matrix_register(ctx.raw(), row_ptr.data(), col.data(), coeff.empty() ? nullptr : coeff[0].l, rows.size(), col.size(), n_vars,
                                           CZK_MEM_HOST, &h_));
    }
    // the same from CSR arrays (row_ptr: rows + 1 offsets, col / coeff: nnz entries), for callers that already hold the matrix flat
    ConstraintMatrix(const Context& ctx, const uint64_t* row_ptr, const uint32_t* col, const Fr* coeff, size_t rows, size_t nnz, size_t n_vars)
        : ctx_(&ctx), rows_(rows) {
        ctx.check(czk_r1cs_matrix_register(ctx.raw(), row_ptr, col, nnz ? coeff->l : nullptr, rows, nnz, n_vars, CZK_MEM_HOST, &h_));
    }
    ConstraintMatrix(const ConstraintMatrix&) = delete;
    ConstraintMatrix& operator=(const ConstraintMatrix&) = delete;
    ~ConstraintMatrix() { czk_r1cs_matrix_release(h_); }
    // rows x full_assignment; the result has `out_len` >= rows elements, the tail zero (`vec![zero; domain_size]`, :66-67)
    std::vector<Fr> evaluate(const std::vector<Fr>& full_assignment, size_t out_len) const {
        std::vector<Fr> out(out_len < rows_ ? rows_ : out_len, Fr{{0, 0, 0, 0}});
        ctx_->check(czk_r1cs_matvec(ctx_->raw(), h_, full_assignment.empty() ? nullptr : full_assignment[0].l, full_assignment.size(), 1,
                                    out[0].l, out.size(), CZK_MEM_HOST));
        return out;
    }
    // the same over share lanes on the GPU: z = lanes x z.capacity() full-assignment lanes, out[lane][0..rows) written, the
    // rest of `out` (the zero padding up to the domain size) untouched; out.len = rows afterwards
    void evaluate(const DeviceLanes& z, DeviceLanes& out) const {
        ctx_->check(czk_r1cs_matvec(ctx_->raw(), h_, z.data(), z.capacity(), z.lanes(), out.data(), out.capacity(), CZK_MEM_DEVICE));
        out.len = rows_;
    }
};

// mpc-snarks/src/groth/r1cs_to_qap.rs:47-113 -- the NTT / pointwise sequence of witness_map for a single prover
// (T = Fr).  `a`, `b`, `c` are the evaluated constraint rows (a[0..N), then the instance copy; :67-83, :95-100).
// `batch_product` is F::batch_product_in_place (:92) -- a plain product here, the Beaver protocol for shares.
struct R1CStoQAP {
    static std::vector<Fr> witness_map(const Context& ctx, const Radix2EvaluationDomain& domain, const std::vector<Fr>& a, const std::vector<Fr>& b,
                                       const std::vector<Fr>& c) {
        // the three vectors go up once, h comes down once: the seven transforms and the pointwise steps run on the resident lanes
        const size_t D = domain.size();
        if (a.size() > D || b.size() > D || c.size() > D) throw Panic(CZK_ERR_SIZE, "assertion failed: coeffs.len() <= self.size()");
        DeviceLanes da(ctx, 1, D), db(ctx, 1, D), dc(ctx, 1, D), dab(ctx, 1, D);
        da.upload(0, a); da.len = a.size();
        db.upload(0, b); db.len = b.size();
        dc.upload(0, c); dc.len = c.size();
        witness_map(domain, da, db, dc, dab, [&](DeviceLanes& x, DeviceLanes& y, DeviceLanes& xy) {
            ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_MUL, x.data(), y.data(), xy.data(), D, CZK_MEM_DEVICE));   // T = Fr: a plain product (:92)
        });
        return dab.to_host(0);
    }

    // T = share lanes resident on the GPU (DeviceLanes of capacity domain.size(); a.len / b.len / c.len = evaluated rows):
    // the same sequence with the fused entry points -- czk_witness_map_pre = :85-89, batch_product = :92 (the caller's Beaver
    // protocol writes a (*) b into `ab`), czk_witness_map_post = :102-110.  `ab` = h on return; nothing leaves HBM.
    template <class BatchProduct>
    static void witness_map(const Radix2EvaluationDomain& domain, DeviceLanes& a, DeviceLanes& b, DeviceLanes& c, DeviceLanes& ab,
                            BatchProduct&& batch_product) {
        const Context& ctx = domain.ctx();
        const size_t lanes = a.lanes();
        if (a.capacity() != domain.size() || b.capacity() != domain.size() || c.capacity() != domain.size() || ab.capacity() != domain.size() ||
            b.lanes() != lanes || c.lanes() != lanes || ab.lanes() != lanes)
            throw Panic(CZK_ERR_SIZE, "witness_map: lanes must have the domain's size");
        ctx.check(czk_witness_map_pre(ctx.raw(), a.data(), a.len, b.data(), b.len, domain.log_size_of_group, lanes));
        a.len = b.len = domain.size();
        batch_product(a, b, ab);
        ctx.check(czk_witness_map_post(ctx.raw(), ab.data(), c.data(), c.len, domain.log_size_of_group, lanes));
        ab.len = c.len = domain.size();
    }
};

}  // namespace czk

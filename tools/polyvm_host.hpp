// polyvm_host.hpp -- a torch-free, Python-free host of the LOCAL compute of the reference's polynomial-IOP provers (Plonk, Marlin), written
// against include/czk.h / czk.hpp only: what the reference's Rust provers do once they link the shim, here in C++.
//
//   reference                                                                                   here
//   mpc-plonk/src/lib.rs:110-258 prove_unit_product / prove_wiring, :259-340 prove_public / prove_gates, :343-448 eval / commit / prove     pvm::plonk_prove
//   marlin/src/ahp/prover.rs:300-704 (three AHP rounds), marlin/src/lib.rs:176-318 (commitments, evaluations, batched openings)             pvm::marlin_prove
//   KZG10::commit / open (poly-commit/src/kzg10/mod.rs:141-265), marlin_pc (poly-commit/src/marlin/marlin_pc/mod.rs:213-330)                Machine::commit / open_*
//   DensePolynomial arithmetic (algebra/poly/src/polynomial/univariate/dense.rs), EvaluationDomain transforms                            Machine::{poly_mul, div_vanishing, ntt, ...}
//
// It is the C++ twin of collaborative-zksnark_amd/polyvm.py (GpuBackend + plonk_prove + marlin_prove), statement for statement: the same
// synthetic circuit / index and SRS from the same seeds, the same fixed stand-ins for Fiat-Shamir challenges (SHA-256(tag) mod r) and blinding
// factors, the same settling of commitments and evaluations at every point where the reference's transcript draws a challenge.  So the two
// hosts must produce the same commitments, evaluations and opening proofs (tests/test_pipelines.py compares them element by element; the Python
// host is itself compared with an oracle-backed machine and pinned to the reference's source by tests/golden/plonk_marlin_callsites.json).
// Every array lives in ONE device arena (a czk_lanes handle) managed by a first-fit free list on the host: no allocation call, no copy and no
// kernel that is not the library's own runs inside a proof -- re-layouts are czk_fr_copy_3d, scalars travel with the launch
// (CZK_MEM_SCALAR_HOST).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "czk.hpp"
#include "groth16_host.hpp"

namespace pvm {

using czk::Fr;
namespace hf = g16::hostfr;

// ---- host field helpers: everything here is in Montgomery form (the library's argument form) ----------------------------------------
inline Fr fr_u64(uint64_t v) { return hf::from_repr(Fr{{v, 0, 0, 0}}); }
inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
inline Fr fr_one() { return hf::one(); }
inline Fr fr_mul(const Fr& a, const Fr& b) { return hf::mont_mul(a, b); }
inline Fr fr_add(const Fr& a, const Fr& b) { return hf::add(a, b); }
inline Fr fr_sub(const Fr& a, const Fr& b) { return hf::sub(a, b); }
inline Fr fr_neg(const Fr& a) { return hf::sub(fr_zero(), a); }
inline bool fr_eq(const Fr& a, const Fr& b) { return memcmp(a.l, b.l, 32) == 0; }
inline Fr fr_pow(const Fr& base, const uint64_t* e, int limbs) {
    Fr acc = fr_one();
    for (int i = 64 * limbs - 1; i >= 0; i--) {
        acc = fr_mul(acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = fr_mul(acc, base);
    }
    return acc;
}
inline Fr fr_pow(const Fr& base, uint64_t e) { return fr_pow(base, &e, 1); }
inline Fr fr_inv(const Fr& a) {   // a^(r - 2)
    uint64_t e[4] = {hf::R_MOD[0] - 2, hf::R_MOD[1], hf::R_MOD[2], hf::R_MOD[3]};
    return fr_pow(a, e, 4);
}
// polyvm.challenge: SHA-256(tag) read as a little-endian integer, mod r
inline Fr challenge(const std::string& tag) {
    uint8_t h[32];
    czk_sha256(tag.data(), tag.size(), h);
    Fr v;
    memcpy(v.l, h, 32);
    while (hf::geq_r(v.l)) hf::sub_r(v.l);   // 2^256 < 8 r: at most seven subtractions
    return hf::from_repr(v);
}
// evaluate_vanishing_polynomial: point^size - 1
inline Fr vanishing(uint64_t size, const Fr& point) { return fr_sub(fr_pow(point, size), fr_one()); }
inline size_t next_pow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
static const int FFT = CZK_FFT, IFFT = CZK_IFFT, COSET_FFT = CZK_COSET_FFT, COSET_IFFT = CZK_COSET_IFFT;

// ---- device arena ------------------------------------------------------------------------------------------------------------------
class Arena {
  public:
    Arena(const czk::Context& ctx, size_t elems) : lanes_(ctx, 1, elems) { free_[0] = elems; }
    // first fit; blocks are multiples of 8 Fr (256 bytes).  Work on the blocks is stream-ordered on the machine's context, so a block
    // released by the host may be handed out again at once: whatever was enqueued on it runs before whatever is enqueued next.
    size_t take(size_t elems) {
        const size_t need = (std::max<size_t>(elems, 1) + 7) & ~(size_t)7;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second >= need) {
                const size_t at = it->first, rest = it->second - need;
                free_.erase(it);
                if (rest) free_[at + need] = rest;
                used_ += need;
                peak_ = std::max(peak_, used_);
                return at;
            }
        }
        throw czk::Panic(CZK_ERR_NOMEM, "pvm::Arena: out of arena memory (raise the arena size)");
    }
    void give(size_t at, size_t elems) {
        size_t need = (std::max<size_t>(elems, 1) + 7) & ~(size_t)7;
        used_ -= need;
        auto nx = free_.lower_bound(at);
        if (nx != free_.end() && at + need == nx->first) {
            need += nx->second;
            nx = free_.erase(nx);
        }
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == at) {
                pv->second += need;
                return;
            }
        }
        free_[at] = need;
    }
    uint64_t* ptr(size_t at) const { return lanes_.data(0, at); }
    size_t peak_elems() const { return peak_; }

  private:
    czk::DeviceLanes lanes_;
    std::map<size_t, size_t> free_;
    size_t used_ = 0, peak_ = 0;
};

struct Block {
    Arena* arena;
    size_t at, elems;
    Block(Arena* a, size_t e) : arena(a), at(a->take(e)), elems(e) {}
    ~Block() { arena->give(at, elems); }
    Block(const Block&) = delete;
    Block& operator=(const Block&) = delete;
};
// a dense lanes x n array of Fr (Montgomery).  Public data has lanes == 1.
struct Arr {
    std::shared_ptr<Block> b;
    size_t lanes = 0, n = 0;
    uint64_t* p() const { return b->arena->ptr(b->at); }
    explicit operator bool() const { return (bool)b; }
};

// ---- results ----------------------------------------------------------------------------------------------------------------------
struct Commitment {   // per lane an affine G1 point (KZG commitments; a hiding commitment is the sum of two MSM results)
    std::vector<czk::G1Projective> jac, jac2;   // jac2: the commitment of the blinding polynomial, added on the host when settled
    std::vector<uint64_t> aff;                  // lanes x 12, valid once settled
    std::vector<uint8_t> inf;
    Arr keep, keep2;                            // the scalars stay referenced until the MSMs have read them
    bool settled = false;
};
using CommitmentP = std::shared_ptr<Commitment>;
struct Value {        // per lane one Fr: an evaluation
    size_t slot = 0, lanes = 0;
    std::vector<Fr> v;
    bool settled = false;
    bool open = false;   // the evaluation of a share polynomial: publicized between the parties (`y.publicize()`, mpc-plonk/src/lib.rs:362-365; marlin/src/lib.rs:283-292)
};
using ValueP = std::shared_ptr<Value>;
struct Opening {
    ValueP value;
    Fr point{};
    CommitmentP proof;
    std::string of;                            // label of the opened polynomial's commitment ("" = an index polynomial committed at setup)
    bool has_of = false;
    ValueP random_v;                           // blinding evaluation of a hiding opening
    std::vector<std::pair<Fr, std::string>> terms;
    Arr wit;                                   // between open_begin and open_finish
};
struct Output {       // everything a prover publishes, in the reference's order
    std::vector<std::vector<Fr>> opened;       // party layout: per transcript point the batch of evaluations opened over the communicator
    std::vector<std::pair<std::string, CommitmentP>> commitments;
    std::vector<std::pair<std::string, Opening>> openings;
    std::vector<std::pair<std::string, std::vector<ValueP>>> evals;
};

// A transcript point in two halves (Machine::mark / Machine::settle): `mark` names what the reference's transcript has absorbed so far -- the commitments and
// evaluations enqueued up to here; `settle` waits for exactly that (czk_ctx_wait_mark) before the next challenge is drawn.  Whatever the prover enqueues between the
// two -- work that depends on no pending challenge -- runs through the wait.
struct Mark {
    uint64_t id = 0;
    std::vector<CommitmentP> cmts;
    std::vector<ValueP> vals;
    std::vector<Fr> host;     // the evaluation rows [row0, row0 + host.size()) on their way down (czk_lanes_download_deferred)
    size_t row0 = 0;
    bool settled = false;
};
using MarkP = std::shared_ptr<Mark>;

// ---- the polynomial machine (polyvm.GpuBackend) -----------------------------------------------------------------------------------
struct Srs {          // powers_of_g = [tau^i] G and powers_of_gamma_g = [gamma tau^i] G for FIXED, KNOWN tau / gamma (polyvm.py GpuBackend.__init__)
    std::unique_ptr<czk::G1Bases> g, gamma;
    size_t n = 0;
};

class Machine {
  public:
    const czk::Context& ctx;
    size_t lanes;
    std::vector<int> lift;     // per lane: 1 = this lane takes public addends (GSZ: every lane; SPDZ: the king's sh and mac lanes)
    std::shared_ptr<Srs> srs;
    size_t msm_count = 0, ntt_count = 0, msm_points = 0;
    // one party per process (the reference's own layout): every batch of evaluations made between two challenges is opened over the communicator as
    // ONE batch_open -- GszFieldShare::batch_open (lanes = 1) or SpdzFieldShare::batch_open (lanes = 2: sh, mac) -- on values that stay in HBM
    const czk::Net* net = nullptr;
    bool net_gsz = false;
    bool commit_opens = true;                  // SPDZ: dx_ts through Net::atomic_broadcast, as the reference does (share/spdz.rs:179, channel.rs:50-75)
    unsigned gsz_degree = 0;
    Fr mac_share{};
    std::vector<std::vector<Fr>> opened;       // what the opens returned, in order (reset by the caller per proof)
    std::vector<std::chrono::steady_clock::time_point>* settle_log = nullptr;   // when each transcript point's wait returned (tools/host_demo.cpp --breakdown)

    Machine(const czk::Context& c, size_t lanes_, size_t max_degree, std::vector<int> lift_, size_t arena_elems, std::shared_ptr<Srs> share = nullptr,
            uint64_t base_seed = 0xBA5E5 + 77)
        : ctx(c), lanes(lanes_), lift(std::move(lift_)), arena_(c, arena_elems), vals_(c, 1, 8192), stage_(c, 1, (size_t)1 << 19), down_(c, 1, 8192) {
        if (lift.empty()) lift.assign(lanes, 1);
        const size_t n = max_degree + 1;
        char tag[64];
        if (share) {
            if (share->n != n) throw czk::Panic(CZK_ERR_ARG, "shared SRS of another size");
            srs = share;
            return;
        }
        srs = std::make_shared<Srs>();
        srs->n = n;
        snprintf(tag, sizeof tag, "kzg.tau.%llx", (unsigned long long)base_seed);
        const Fr tau = challenge(tag);
        snprintf(tag, sizeof tag, "kzg.gamma.%llx", (unsigned long long)base_seed);
        const Fr gamma = challenge(tag);
        {
            Arr pw = alloc(1, n), k = alloc(1, n);
            ctx.check(czk_fr_powers(ctx.raw(), tau.l, nullptr, n, pw.p(), CZK_MEM_DEVICE));
            ctx.check(czk_fr_into_repr(ctx.raw(), pw.p(), k.p(), n, CZK_MEM_DEVICE));
            czk::DeviceLanes pts(ctx, 1, 3 * n);   // n x 12 u64 = 3 n Fr-sized slots
            ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, k.p(), n, pts.data(), CZK_MEM_DEVICE));
            srs->g.reset(new czk::G1Bases(ctx, pts.data(), nullptr, n, CZK_MEM_DEVICE));
            ctx.sync();
        }
        {
            std::vector<Fr> kc(8);
            Fr t = gamma;
            for (int i = 0; i < 8; i++, t = fr_mul(t, tau)) kc[i] = t;
            ctx.check(czk_fr_into_repr(ctx.raw(), kc[0].l, kc[0].l, 8, CZK_MEM_HOST));
            std::vector<uint64_t> gp(8 * 12);
            ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, kc[0].l, 8, gp.data(), CZK_MEM_HOST));
            srs->gamma.reset(new czk::G1Bases(ctx, gp.data(), nullptr, 8));
        }
    }
    void prepare(const std::vector<size_t>& sizes) {   // czk_bases_prepare for the commitment lengths a prover uses (at SRS load, not inside the first proof)
        std::vector<size_t> s(sizes);
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        for (size_t n : s)
            if (n > 0 && n <= srs->n) ctx.check(czk_bases_prepare(ctx.raw(), srs->g->raw(), n));
    }
    size_t arena_peak_bytes() const { return arena_.peak_elems() * 32; }

    // ---- storage and re-layouts: czk_fr_copy_3d on the context's stream --------------------------------------------------------------
    Arr alloc(size_t ln, size_t n) {
        Arr a;
        a.b = std::make_shared<Block>(&arena_, ln * n);
        a.lanes = ln, a.n = n;
        return a;
    }
    void copy(const Arr& dst, size_t dst_off, const size_t (&ds)[3], const Arr* src, size_t src_off, const size_t (&ss)[3], const size_t (&n3)[3]) {
        if (n3[0] * n3[1] * n3[2] == 0) return;
        ctx.check(czk_fr_copy_3d(ctx.raw(), dst.p() + 4 * dst_off, ds, src ? src->p() + 4 * src_off : nullptr, ss, n3));
    }
    Arr zeros(size_t ln, size_t n) {
        Arr o = alloc(ln, n);
        copy(o, 0, {0, n, 1}, nullptr, 0, {0, 0, 0}, {1, ln, n});
        return o;
    }
    Arr upload(const std::vector<Fr>& v, size_t ln, size_t n) {
        // host values -> arena, through the machine's staging handle in pieces: czk_lanes_upload returns once the host vector has been read, the DMA
        // and the copy out of the staging area run in stream order -- no synchronisation, no allocation (a proof uploads a two-element vector; inputs
        // go up at setup)
        Arr o = alloc(ln, n);
        const size_t cap = stage_.capacity(), st[3] = {0, 0, 1};
        for (size_t at = 0; at < ln * n; at += cap) {
            const size_t m = std::min(cap, ln * n - at), n3[3] = {1, 1, m};
            stage_.upload(0, 0, v.data() + at, m);
            ctx.check(czk_fr_copy_3d(ctx.raw(), o.p() + 4 * at, st, stage_.data(), st, n3));
        }
        return o;
    }
    Arr lane_stack(const std::vector<Arr>& parts) {
        size_t tot = 0;
        for (const Arr& p : parts) tot += p.lanes;
        const size_t n = parts[0].n;
        Arr o = alloc(tot, n);
        size_t at = 0;
        for (const Arr& p : parts) {
            copy(o, at * n, {0, n, 1}, &p, 0, {0, n, 1}, {1, p.lanes, n});
            at += p.lanes;
        }
        return o;
    }
    Arr lifted(const Arr& a_public) {   // the public array on the lifting lanes, zero elsewhere: runs of equal lanes are one copy (source lane stride 0) or one fill
        const size_t n = a_public.n;
        Arr o = alloc(lanes, n);
        for (size_t at = 0; at < lanes;) {
            size_t end = at;
            while (end < lanes && lift[end] == lift[at]) end++;
            copy(o, at * n, {0, n, 1}, lift[at] ? &a_public : nullptr, 0, {0, 0, 1}, {1, end - at, n});
            at = end;
        }
        return o;
    }
    Arr shared_copy(const Arr& a_public) { return lane_stack(std::vector<Arr>(lanes, a_public)); }
    Arr resized(const Arr& a, size_t n) {
        if (n == a.n) return a;   // arrays are never written in place: the same array serves
        Arr o = alloc(a.lanes, n);
        const size_t m = std::min(n, a.n);
        copy(o, 0, {0, n, 1}, &a, 0, {0, a.n, 1}, {1, a.lanes, m});
        copy(o, m, {0, n, 1}, nullptr, 0, {0, 0, 0}, {1, a.lanes, n - m});
        return o;
    }
    Arr drop_first(const Arr& a, size_t k) {
        Arr o = alloc(a.lanes, a.n - k);
        copy(o, 0, {0, a.n - k, 1}, &a, k, {0, a.n, 1}, {1, a.lanes, a.n - k});
        return o;
    }
    Arr concat(const std::vector<Arr>& parts) {
        size_t tot = 0;
        for (const Arr& p : parts) tot += p.n;
        Arr o = alloc(parts[0].lanes, tot);
        size_t at = 0;
        for (const Arr& p : parts) {
            copy(o, at, {0, tot, 1}, &p, 0, {0, p.n, 1}, {1, p.lanes, p.n});
            at += p.n;
        }
        return o;
    }
    Arr strided_split(const Arr& a, size_t n) {   // out[l n + j][k] = a[l][k n + j]
        const size_t L = a.n / n;
        Arr o = alloc(a.lanes * n, L);
        copy(o, 0, {n * L, L, 1}, &a, 0, {a.n, 1, n}, {a.lanes, n, L});
        return o;
    }
    Arr strided_merge(const Arr& a, size_t n, size_t ln) {   // out[l][k n + j] = a[l n + j][k]
        const size_t L = a.n;
        Arr o = alloc(ln, L * n);
        copy(o, 0, {L * n, n, 1}, &a, 0, {n * L, 1, L}, {ln, L, n});
        return o;
    }
    // ---- arithmetic -------------------------------------------------------------------------------------------------------------------
    Arr vec(int op, Arr a, Arr b) {
        if (op == CZK_OP_MUL && a.lanes != b.lanes) {   // a public operand is broadcast over the lanes
            Arr& one = a.lanes == 1 ? a : b;
            const size_t many = a.lanes == 1 ? b.lanes : a.lanes;
            Arr rep = alloc(many, one.n);
            copy(rep, 0, {0, one.n, 1}, &one, 0, {0, 0, 1}, {1, many, one.n});
            one = rep;
        }
        if (a.lanes != b.lanes || a.n != b.n) throw czk::Panic(CZK_ERR_ARG, "pvm: shape mismatch");
        Arr o = alloc(a.lanes, a.n);
        ctx.check(czk_fr_vec_op(ctx.raw(), op, a.p(), b.p(), o.p(), a.lanes * a.n, CZK_MEM_DEVICE));
        return o;
    }
    Arr add(const Arr& a, const Arr& b) { return vec(CZK_OP_ADD, a, b); }
    Arr sub(const Arr& a, const Arr& b) { return vec(CZK_OP_SUB, a, b); }
    Arr mul(const Arr& a, const Arr& b) { return vec(CZK_OP_MUL, a, b); }
    // public +- shared is the reference's `shift`: the public operand is added on the lifting lanes only
    Arr plus(const Arr& a, const Arr& b) { return a.lanes == b.lanes ? add(a, b) : a.lanes == 1 ? add(lifted(a), b) : add(a, lifted(b)); }
    Arr minus(const Arr& a, const Arr& b) { return a.lanes == b.lanes ? sub(a, b) : a.lanes == 1 ? sub(lifted(a), b) : sub(a, lifted(b)); }
    Arr scale(const Arr& a, const Fr& k) {
        Arr o = alloc(a.lanes, a.n);
        ctx.check(czk_fr_vec_scale(ctx.raw(), a.p(), k.l, o.p(), a.lanes * a.n, CZK_MEM_DEVICE | CZK_MEM_SCALAR_HOST));
        return o;
    }
    Arr powers(const Fr& g, size_t n) {
        Arr o = alloc(1, n);
        ctx.check(czk_fr_powers(ctx.raw(), g.l, nullptr, n, o.p(), CZK_MEM_DEVICE));
        return o;
    }
    Arr constant(const Fr& k, size_t n) {   // n copies of k: k * 1^i
        Arr o = alloc(1, n);
        const Fr one = fr_one();
        ctx.check(czk_fr_powers(ctx.raw(), one.l, k.l, n, o.p(), CZK_MEM_DEVICE));
        return o;
    }
    Arr add_const(const Arr& a, const Fr& k) { return plus(a, constant(k, a.n)); }   // k on every element: arrays of EVALUATIONS
    // `&DensePolynomial + &F` (algebra/poly/src/polynomial/univariate/dense.rs:301-315): a polynomial in COEFFICIENT form takes a constant on coefficient 0 only
    Arr poly_add_const(const Arr& a, const Fr& k) { return plus(a, resized(constant(k, 1), a.n)); }
    Arr ntt(const Arr& a, size_t size, int kind) {
        const size_t m = std::min(a.n, size);
        Arr o = alloc(a.lanes, size);
        if ((size & (size - 1)) == 0 && m > 0) {   // radix-2: the transform reads the source lanes itself (EvaluationDomain::fft(&coeffs) -> Vec)
            unsigned log_d = 0;
            while (((size_t)1 << log_d) < size) log_d++;
            ctx.check(czk_ntt_fr_to(ctx.raw(), a.p(), a.n, o.p(), log_d, a.lanes, kind, m, CZK_MEM_DEVICE));
        } else {
            copy(o, 0, {0, size, 1}, &a, 0, {0, a.n, 1}, {1, a.lanes, m});
            ctx.check(czk_ntt_fr_mixed(ctx.raw(), o.p(), size, a.lanes, kind, m, CZK_MEM_DEVICE));
        }
        ntt_count += a.lanes;
        return o;
    }
    Arr prefix_product(const Arr& a) {
        Arr o = alloc(a.lanes, a.n);
        for (size_t ln = 0; ln < a.lanes; ln++) ctx.check(czk_fr_prefix_product(ctx.raw(), a.p() + 4 * ln * a.n, a.n, o.p() + 4 * ln * a.n, CZK_MEM_DEVICE));
        return o;
    }
    Arr inverse(const Arr& a) {
        Arr o = alloc(a.lanes, a.n);
        ctx.check(czk_fr_batch_inverse(ctx.raw(), a.p(), a.lanes * a.n, nullptr, o.p(), CZK_MEM_DEVICE));
        return o;
    }
    Arr random(uint64_t seed, size_t n) {   // rand_fr_canonical(seed, n) in Montgomery form, public
        std::vector<Fr> c = g16::rand_fr_canonical(seed, n);
        Arr o = upload(c, 1, n);
        ctx.check(czk_fr_from_repr(ctx.raw(), o.p(), o.p(), n, CZK_MEM_DEVICE));
        return o;
    }
    Fr root_of_unity(size_t size) {
        uint64_t k[24];
        ctx.check(czk_mixed_domain_constants(ctx.raw(), size, k));
        return Fr{{k[4], k[5], k[6], k[7]}};   // group_gen
    }
    Arr shift(const Arr& a, const Fr& w) { return mul(a, powers(w, a.n)); }   // mpc-plonk/src/util.rs:11-18: coefficient i times w^i
    // `&DensePolynomial * &DensePolynomial`: both operands over GeneralEvaluationDomain::new(len a + len b - 1) (always radix-2), point-wise, interpolated
    Arr poly_mul(const Arr& a, const Arr& b) {
        const size_t n = a.n + b.n - 1, size = next_pow2(n);
        return resized(ntt(mul(ntt(a, size, FFT), ntt(b, size, FFT)), size, IFFT), n);
    }
    // a b c over ONE domain: the reference forms `&a * &(&b * &c)` as two products, interpolating b c and transforming it again -- on the SAME domain whenever
    // next_pow2(len b + len c - 1) == next_pow2(len a + len b + len c - 2); the inverse transform followed by the forward one is then the identity on
    // exact field elements, so multiplying the three evaluation vectors gives the same coefficients with two transforms fewer per lane
    Arr poly_mul3(const Arr& a, const Arr& b, const Arr& c) {
        const size_t n = a.n + b.n + c.n - 2, size = next_pow2(n);
        if (next_pow2(b.n + c.n - 1) != size) return poly_mul(a, poly_mul(b, c));
        return resized(ntt(mul(ntt(a, size, FFT), mul(ntt(b, size, FFT), ntt(c, size, FFT))), size, IFFT), n);
    }
    // out[i] = a[(i + k) mod n] per lane: the evaluations of f(w^k X) over a domain (or a coset of it) generated by w are those of f, k places on
    Arr rotated(const Arr& a, size_t k) {
        k %= a.n;
        if (k == 0) return a;
        Arr o = alloc(a.lanes, a.n);
        copy(o, 0, {0, a.n, 1}, &a, k, {0, a.n, 1}, {1, a.lanes, a.n - k});
        copy(o, a.n - k, {0, a.n, 1}, &a, 0, {0, a.n, 1}, {1, a.lanes, k});
        return o;
    }
    Arr padded_add(const Arr& a, const Arr& b) {
        const size_t n = std::max(a.n, b.n);
        return plus(resized(a, n), resized(b, n));
    }
    // ---- values: every evaluation made between two transcript points lands in one device array ---------------------------------------------
    // pub: 1 = the polynomial is public data (its value needs no opening between the parties), 0 = a share polynomial, -1 = infer from the lane count
    bool value_is_open(const Arr& a, int pub) const { return !(pub < 0 ? (a.lanes == 1 && lanes > 1) : pub != 0); }
    ValueP value_slot(size_t rows, bool open = false) {
        if (vals_at_ + rows > vals_.capacity()) throw czk::Panic(CZK_ERR_SIZE, "pvm: evaluation buffer full");
        ValueP v = std::make_shared<Value>();
        v->slot = vals_at_, v->lanes = rows, v->open = open;
        vals_at_ += rows;
        pending_vals_.push_back(v);
        return v;
    }
    // a / (X - z): quotient and remainder (KZG10::compute_witness_polynomial, poly-commit/src/kzg10/mod.rs:200-224)
    std::pair<Arr, ValueP> div_linear(const Arr& a, const Fr& z, int pub = 1) {
        Arr q = alloc(a.lanes, a.n ? a.n - 1 : 0);
        ValueP rem = value_slot(a.lanes, value_is_open(a, pub));
        ctx.check(czk_poly_div_linear(ctx.raw(), a.p(), a.n, a.lanes, z.l, q.p(), vals_.data(0, rem->slot), CZK_MEM_DEVICE));
        return {q, rem};
    }
    Arr quotient(const Arr& a, const Fr& x) { return div_linear(a, x).first; }
    // Polynomial::evaluate per lane, without the quotient.  The evaluations of a round are collected and go to the GPU together (czk_poly_evaluate_many, one
    // launch per reduction level for all of them) when the round's transcript point is marked; the polynomial stays referenced until then.
    ValueP evaluate(const Arr& a, const Fr& x, int pub = -1) {
        ValueP v = value_slot(a.lanes, value_is_open(a, pub));
        eval_queue_.push_back(EvalReq{a, x, v->slot});
        return v;
    }
    void flush_evaluations() {
        if (eval_queue_.empty()) return;
        std::vector<const uint64_t*> src;
        std::vector<uint64_t*> dst;
        std::vector<size_t> n, ln;
        std::vector<Fr> z;
        for (const EvalReq& r : eval_queue_) {
            src.push_back(r.a.p()), dst.push_back(vals_.data(0, r.slot)), n.push_back(r.a.n), ln.push_back(r.a.lanes), z.push_back(r.x);
        }
        ctx.check(czk_poly_evaluate_many(ctx.raw(), src.size(), src.data(), n.data(), ln.data(), z[0].l, dst.data()));
        eval_queue_.clear();
    }
    // sum_k c_k a_k over arrays of different lengths, public terms on the lifting lanes only (`plus`): `poly += (coeff, &other)` per term, one pass
    // (czk_fr_lincomb) instead of a scale, two resizes and an addition per term
    Arr lincomb(const std::vector<std::pair<Fr, Arr>>& terms, const Fr* constant = nullptr) {
        size_t ln = 1, len = 0;
        for (auto& t : terms) ln = std::max(ln, t.second.lanes), len = std::max(len, t.second.n);
        uint64_t mask = 0;
        for (size_t l = 0; l < lanes && l < 64; l++) mask |= (uint64_t)(lift[l] ? 1 : 0) << l;
        Arr acc;
        for (size_t at = 0; at < terms.size();) {
            std::vector<const uint64_t*> src;
            std::vector<size_t> n, tl;
            std::vector<Fr> c;
            if (acc) src.push_back(acc.p()), n.push_back(acc.n), tl.push_back(acc.lanes), c.push_back(fr_one());
            for (; at < terms.size() && src.size() < 12; at++) {
                const Arr& a = terms[at].second;
                if (a.lanes != ln && a.lanes != 1) throw czk::Panic(CZK_ERR_ARG, "pvm: lincomb of arrays with different lane counts");
                src.push_back(a.p()), n.push_back(a.n), tl.push_back(a.lanes), c.push_back(terms[at].first);
            }
            Arr o = alloc(ln, len);
            ctx.check(czk_fr_lincomb(ctx.raw(), src.size(), src.data(), n.data(), tl.data(), c[0].l, (constant && !acc) ? constant->l : nullptr, ln, mask, o.p(), len));
            acc = o;
        }
        return acc;
    }
    // `divide_by_vanishing_poly` in coefficient form: (q, r) with a = q (X^n - 1) + r
    std::pair<Arr, Arr> div_vanishing(const Arr& a, size_t n) {
        const size_t m = a.n;
        if (m <= n) return {resized(a, 0), a};
        if (n < (m + n - 1) / n) {
            // few residues, many terms (Marlin divides by v_X with |X| = 2): the n residue classes are independent polynomials in Y = X^n, and q's
            // class c is the quotient of that polynomial by (Y - 1)
            const size_t L = (m + n - 1) / n;
            Arr classes = strided_split(resized(a, L * n), n);   // (lanes n, L)
            Arr qc = alloc(classes.lanes, L - 1), r = alloc(a.lanes, n);
            ctx.check(czk_poly_div_linear(ctx.raw(), classes.p(), L, classes.lanes, fr_one().l, qc.p(), r.p(), CZK_MEM_DEVICE));   // remainders: lanes x n, in place
            Arr q = strided_merge(resized(qc, L), n, a.lanes);
            return {resized(q, m - n), r};
        }
        // q_i = sum_{k >= 1} a_{i + k n}, r_i = sum_{k >= 0} a_{i + k n}: the suffix sums of a's n-coefficient chunks, one pass (czk_poly_div_vanishing)
        Arr q = alloc(a.lanes, m - n), r = alloc(a.lanes, n);
        ctx.check(czk_poly_div_vanishing(ctx.raw(), a.p(), m, a.lanes, n, q.p(), r.p(), CZK_MEM_DEVICE));
        return {q, r};
    }
    // ---- commitments and openings -------------------------------------------------------------------------------------------------------
    CommitmentP commit(const Arr& a, bool gamma_key = false) {   // KZG10::commit's MSM, enqueued (czk_msm_async); transcript_point() settles it
        czk::G1Bases& bases = gamma_key ? *srs->gamma : *srs->g;
        if (a.n > bases.len()) throw czk::Panic(CZK_ERR_SIZE, "polynomial longer than the committer key");
        CommitmentP c = std::make_shared<Commitment>();
        c->jac.resize(a.lanes);
        c->keep = a;
        // CZK_MEM_STABLE: `keep` holds the scalars until the commitment is settled, so the context's stream need not wait for the digit extraction
        ctx.check(czk_msm_async(ctx.raw(), bases.raw(), a.p(), a.n, a.lanes, CZK_SCALAR_MONTGOMERY, CZK_MEM_DEVICE | CZK_MEM_STABLE, c->jac[0].x.l));
        msm_count += a.lanes;
        msm_points += a.lanes * a.n;
        pending_cmts_.push_back(c);
        return c;
    }
    // `commitment.add_assign_mixed(&random_commitment)` (kzg10/mod.rs:188) / `w += ...` (:246-249): the second MSM, added on the host when settled
    CommitmentP commit_sum(CommitmentP c, const Arr& blind) {
        czk::G1Bases& bases = *srs->gamma;
        c->jac2.resize(blind.lanes);
        c->keep2 = blind;
        ctx.check(czk_msm_async(ctx.raw(), bases.raw(), blind.p(), blind.n, blind.lanes, CZK_SCALAR_MONTGOMERY, CZK_MEM_DEVICE | CZK_MEM_STABLE, c->jac2[0].x.l));
        msm_count += blind.lanes;
        msm_points += blind.lanes * blind.n;
        return c;
    }
    Opening open_begin(const Arr& a, const Fr& x, int pub = -1) {
        Opening o;
        auto qr = div_linear(a, x, pub);
        o.wit = qr.first;
        o.value = qr.second;
        o.point = x;
        return o;
    }
    Opening open_finish(Opening o) {
        o.proof = commit(o.wit);
        o.wit = Arr();
        return o;
    }
    Opening open_at(const Arr& a, const Fr& x, int pub = -1) { return open_finish(open_begin(a, x, pub)); }
    // Where the reference feeds commitments / evaluations to its Fiat-Shamir transcript before drawing the next challenge: everything committed
    // or evaluated SO FAR IN THE REFERENCE'S ORDER must be final.  mark() at that place in the sequence, settle() before the challenge is used; a
    // prover with nothing to enqueue in between calls transcript_point().
    MarkP mark() {
        flush_evaluations();
        MarkP m = std::make_shared<Mark>();
        m->cmts.swap(pending_cmts_);
        m->vals.swap(pending_vals_);
        if (vals_at_ > vals_done_) {   // one copy: every slot handed out since the last mark, delivered with the mark
            m->row0 = vals_done_;
            m->host.resize(vals_at_ - vals_done_);
            vals_.download_deferred(0, vals_done_, m->host.data(), m->host.size());
            vals_done_ = vals_at_;
        }
        m->id = ctx.mark();
        marks_.push_back(m);
        if (net) settle(m);   // one party per process: the evaluations are opened over the communicator here, in the reference's batches
        return m;
    }
    void settle(const MarkP& upto) {
        while (!marks_.empty() && !upto->settled) {
            MarkP m = marks_.front();
            marks_.pop_front();
            settle_one(*m);
        }
        if (marks_.empty() && pending_cmts_.empty() && pending_vals_.empty()) vals_at_ = vals_done_ = 0;   // nothing refers to the evaluation buffer any more
    }
    void transcript_point() {
        if (pending_cmts_.empty() && pending_vals_.empty() && marks_.empty()) return;
        settle(mark());
    }

  private:
    void settle_one(Mark& m) {
        ctx.wait_mark(m.id);
        if (settle_log) settle_log->push_back(std::chrono::steady_clock::now());
        // into_affine of every commitment settled here in ONE call (the library shares one field inversion over the array: czk_jac_to_affine); a hiding
        // commitment first adds its blinding commitment: into_affine of both, add_assign_mixed (GroupProjective::add_assign_mixed, kzg10/mod.rs:188), as
        // polyvm.group_add does -- the sums then go through the same call
        auto to_affine = [&](const std::vector<czk::G1Projective>& jac, std::vector<uint64_t>& aff, std::vector<uint8_t>& inf) {
            aff.resize(12 * jac.size());
            inf.resize(jac.size());
            if (!jac.empty()) ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, jac[0].x.l, jac.size(), aff.data(), inf.data()));
        };
        std::vector<czk::G1Projective> blinded;
        std::vector<uint64_t> aff;
        std::vector<uint8_t> inf;
        for (CommitmentP& c : m.cmts)
            if (!c->jac2.empty()) {
                blinded.insert(blinded.end(), c->jac.begin(), c->jac.end());
                blinded.insert(blinded.end(), c->jac2.begin(), c->jac2.end());
            }
        to_affine(blinded, aff, inf);
        size_t at = 0;
        for (CommitmentP& c : m.cmts)
            if (!c->jac2.empty()) {
                const size_t L = c->jac.size();
                for (size_t ln = 0; ln < L; ln++) {
                    czk::G1Projective acc{}, sum;
                    ctx.check(czk_jac_add_mixed(ctx.raw(), CZK_G1, acc.x.l, &aff[12 * (at + ln)], inf[at + ln], acc.x.l));
                    ctx.check(czk_jac_add_mixed(ctx.raw(), CZK_G1, acc.x.l, &aff[12 * (at + L + ln)], inf[at + L + ln], sum.x.l));
                    c->jac[ln] = sum;
                }
                at += 2 * L;
            }
        std::vector<czk::G1Projective> all;
        for (CommitmentP& c : m.cmts) all.insert(all.end(), c->jac.begin(), c->jac.end());
        to_affine(all, aff, inf);
        at = 0;
        for (CommitmentP& c : m.cmts) {
            const size_t L = c->jac.size();
            c->aff.assign(aff.begin() + 12 * at, aff.begin() + 12 * (at + L));
            c->inf.assign(inf.begin() + at, inf.begin() + at + L);
            at += L;
            c->keep = c->keep2 = Arr();
            c->settled = true;
        }
        std::vector<ValueP> to_open;
        for (ValueP& v : m.vals) {
            const Fr* at = m.host.data() + (v->slot - m.row0);
            v->v.assign(at, at + v->lanes);
            v->settled = true;
            if (v->open) to_open.push_back(v);
        }
        if (net && !to_open.empty()) {
            // `y.publicize()` of every evaluation made since the last challenge, as ONE batch_open over the parties: lane j of the k values
            // gathered into k contiguous elements (k is a few dozen), the open on the communicator, the opened vector to the host
            const size_t k = to_open.size();
            Arr sh = alloc(1, k), mac = alloc(1, k), res = alloc(1, k);
            const size_t one3[3] = {1, 1, 1}, st[3] = {0, 0, 1};
            for (size_t i = 0; i < k; i++) {
                if (to_open[i]->lanes != lanes) throw czk::Panic(CZK_ERR_ARG, "pvm: an opened value must have one element per local lane");
                ctx.check(czk_fr_copy_3d(ctx.raw(), sh.p() + 4 * i, st, vals_.data(0, to_open[i]->slot), st, one3));
                if (!net_gsz) ctx.check(czk_fr_copy_3d(ctx.raw(), mac.p() + 4 * i, st, vals_.data(0, to_open[i]->slot + 1), st, one3));
            }
            if (net_gsz) net->gsz_batch_open(sh.p(), k, gsz_degree, res.p());
            else {
                uint64_t bad = 0;
                net->check(czk_spdz_batch_open(net->raw(), sh.p(), mac.p(), mac_share.l, k, res.p(), commit_opens ? CZK_OPEN_COMMIT : 0, &bad));
                if (bad) throw czk::Panic(CZK_ERR_CHECK, "assertion failed: sum.is_zero() (SPDZ MAC check, share/spdz.rs:183)");
            }
            const size_t kk[3] = {1, 1, k};
            ctx.check(czk_fr_copy_3d(ctx.raw(), down_.data(), st, res.p(), st, kk));
            std::vector<Fr> got(k);
            down_.download(0, 0, got.data(), k);
            opened.push_back(got);
        }
        m.cmts.clear();
        m.vals.clear();
        m.settled = true;
    }

    struct EvalReq {
        Arr a;
        Fr x;
        size_t slot;
    };
    std::vector<EvalReq> eval_queue_;
    Arena arena_;
    czk::DeviceLanes vals_, stage_, down_;   // evaluation buffer; upload staging (16 MiB); opened values on their way to the host
    size_t vals_at_ = 0, vals_done_ = 0;   // evaluation rows handed out / already on their way to the host
    std::vector<CommitmentP> pending_cmts_;
    std::vector<ValueP> pending_vals_;
    std::deque<MarkP> marks_;                // taken, not yet settled (oldest first)
};

static const Fr GENERATOR = fr_u64(22);   // Fr::multiplicative_generator() (fr.rs:69-74)

// ====================================================================================================================================
// Plonk (mpc-plonk/src/lib.rs)
// ====================================================================================================================================
struct PlonkInputs {
    size_t n_gates;
    Arr p, s, w;   // wire values (share lanes, 3 n_gates), selector (public, n_gates), wiring permutation (public, 3 n_gates)
};
inline PlonkInputs plonk_inputs(Machine& B, size_t n_gates, uint64_t seed = 0x9107) {
    const size_t W = 3 * n_gates;
    return PlonkInputs{n_gates, B.shared_copy(B.random(seed + 1, W)), B.random(seed + 2, n_gates), B.random(seed + 3, W)};
}
inline size_t plonk_max_degree(size_t n_gates) { return 6 * n_gates; }
inline std::vector<size_t> plonk_commit_sizes(size_t G) { return {3 * G, 3 * G - 1, 6 * G - 2, 6 * G - 3, G - 1}; }

// Local compute of `Prover::prove` (mpc-plonk/src/lib.rs:430-448): every commitment and opening in the reference's order
inline Output plonk_prove(Machine& B, const PlonkInputs& inp) {
    const size_t G = inp.n_gates, W = 3 * G;
    const Fr w = B.root_of_unity(W);                                   // domains.wires.group_gen (mixed radix)
    const Fr zinv_w = fr_inv(vanishing(W, GENERATOR));                 // divide_by_vanishing_poly_on_coset (domain/mod.rs:184-191)
    const Fr ww = fr_mul(w, w);
    Output out;
    const Arr &p = inp.p, &s_pub = inp.s, &w_pub = inp.w;
    auto commit = [&](const char* label, const Arr& a) { out.commitments.emplace_back(std::string(label) + "_cmt", B.commit(a)); };
    auto open_ = [&](const char* label, const Arr& a, const Fr& x, const char* of) {
        Opening o = B.open_at(a, x, of == nullptr ? 1 : 0);   // public = the polynomial is an index polynomial (no commitment of this proof)
        o.has_of = of != nullptr;
        if (of) o.of = of;
        out.openings.emplace_back(label, std::move(o));
    };
    // Schedule: the statements below are the reference's, and every commitment / opening lands in `out` in the reference's order; what moves is WHEN a
    // statement is enqueued.  A transcript point waits for what the transcript has absorbed (B.mark() at that place, B.settle() before the challenge is
    // used); statements that depend on no pending challenge are enqueued between the two and run through the wait, so the accumulate kernels of the
    // commitments being settled share the GPU with the next statements' transforms and the MSM queue does not run dry at the drain.
    commit("p", p);                                                    // :434-441
    // prove_public (:259-292) with one public wire: v = p(x_pub) constant, z = X - x_pub, q = (p - v) / z
    Arr q_pub = B.quotient(p, w);
    commit("pub_q", q_pub);
    MarkP tp = B.mark();                                               // transcript: p, pub_q
    // prove_gates (:295-340): d = s (p + pw) + (1 - s)(p pw) - pww, q = d / v_gates -- no challenge enters.  Its first two products are enqueued ahead of
    // the first challenge (they run under the accumulate kernels of p and pub_q), the rest behind the two openings at that challenge: transforms make slow
    // progress while an accumulate kernel holds the register files, so the openings -- which feed the MSM queue -- must not queue behind all of them
    Arr pw = B.shift(p, w), pww = B.shift(p, ww);
    Arr one_minus_s = B.poly_add_const(B.scale(s_pub, fr_neg(fr_one())), fr_one());   // public: `&(&circ.s * &-F::one()) + &F::one()` (:307-308)
    Arr s_ppw = B.poly_mul(s_pub, B.add(p, pw));
    B.settle(tp);
    Fr x = challenge("plonk.public.x");
    open_("pub_q_open", q_pub, x, "pub_q");
    open_("pub_p_open", p, x, "p");
    // (1 - s) (p pw): `&(&(&circ.s * &-F::one()) + &F::one()) * &(p.polynomial() * &pw)` (:307) on one domain (Machine::poly_mul3: same coefficients, two transforms per lane fewer)
    Arr d = B.sub(B.padded_add(s_ppw, B.poly_mul3(one_minus_s, p, pw)), B.resized(pww, G + 2 * W - 2));
    Arr q_gates = B.div_vanishing(d, G).first;
    pw = pww = one_minus_s = d = s_ppw = Arr();
    commit("gates_q", q_gates);
    tp = B.mark();                                                     // transcript: + the two openings, gates_q
    // prove_wiring's evaluations of p and w over the wire domain (:207-211) depend on no challenge either
    Arr p_evals = B.ntt(p, W, FFT), w_evals = B.ntt(w_pub, W, FFT);
    B.settle(tp);
    x = challenge("plonk.gates.x");
    open_("gates_s_open", s_pub, x, nullptr);
    open_("gates_p_open", p, x, "p");
    open_("gates_q_open", q_gates, x, "gates_q");
    open_("gates_p_w_open", p, fr_mul(w, x), "p");
    open_("gates_p_w2_open", p, fr_mul(ww, x), "p");
    // prove_wiring (:201-257) over the wire domain
    B.transcript_point();
    const Fr y = challenge("plonk.wiring.y"), z = challenge("plonk.wiring.z");
    Arr yx_z = B.ntt(B.upload({z, y}, 1, 2), W, FFT);
    Arr num_evals = B.add_const(B.plus(p_evals, B.scale(w_evals, y)), z);
    Arr den_evals = B.plus(p_evals, yx_z);
    Arr l1_evals = B.mul(num_evals, B.inverse(den_evals));
    Arr l1 = B.ntt(l1_evals, W, IFFT);
    commit("l1", l1);
    // prove_unit_product(l1) (:115-198).  Three transforms of the reference's sequence are identities on exact field elements and are not repeated here:
    //   f.evaluate_over_domain_by_ref (:119) of f = l1 = l1_evals.interpolate() (:217) is l1_evals itself;
    //   the coset evaluations of f(w X) and t(w X) (distribute_powers by w, then coset_fft: :141-143, :151-153) are those of f and t, one place on
    //   (w = domain.element(1) generates the wire domain: f(w g w^i) = f(g w^(i+1)));
    //   and l1's coset evaluations serve the product argument and l2_q (:224) alike.
    Arr t = B.ntt(B.prefix_product(l1_evals), W, IFFT);
    commit("t", t);
    Arr l1_v = B.ntt(l1, W, COSET_FFT), t_c = B.ntt(t, W, COSET_FFT);
    Arr f_c = B.rotated(l1_v, 1), tw_c = B.rotated(t_c, 1);
    Arr q_up = B.ntt(B.scale(B.sub(tw_c, B.mul(f_c, t_c)), zinv_w), W, COSET_IFFT);
    commit("q", q_up);
    tp = B.mark();                                                     // transcript: l1, t, q
    // l2_q (:228-243) depends on y and z only: enqueued ahead of the product argument's challenge r
    Arr num_c = B.ntt(num_evals, W, IFFT), den_c = B.ntt(den_evals, W, IFFT);   // interpolate() of both before the coset transforms (:225-226)
    Arr num_v = B.ntt(num_c, W, COSET_FFT), den_v = B.ntt(den_c, W, COSET_FFT);
    Arr l2_q = B.ntt(B.scale(B.sub(B.mul(l1_v, den_v), num_v), zinv_w), W, COSET_IFFT);
    CommitmentP l2_cmt = B.commit(l2_q);
    B.settle(tp);
    const Fr r = challenge("plonk.product.r");
    open_("t_wr_open", t, fr_mul(w, r), "t");
    open_("t_r_open", t, r, "t");
    open_("t_wk_open", t, fr_pow(w, W - 1), "t");
    open_("f_wr_open", l1, fr_mul(w, r), "l1");
    open_("q_r_open", q_up, r, "q");
    out.commitments.emplace_back("l2_q_cmt", l2_cmt);
    B.transcript_point();
    x = challenge("plonk.wiring.x");
    open_("l2_q_x_open", l2_q, x, "l2_q");
    open_("w_x_open", w_pub, x, nullptr);
    open_("l1_x_open", l1, x, "l1");
    open_("p_x_open", p, x, "p");
    B.transcript_point();
    out.opened = std::move(B.opened);
    B.opened.clear();
    return out;
}

// ====================================================================================================================================
// Marlin (marlin/src/ahp/prover.rs, marlin/src/lib.rs)
// ====================================================================================================================================
struct MarlinStar {
    Arr on_K[3];   // row, col, val
    Arr on_B[4];   // row, col, row_col, val
};
struct MarlinInputs {
    size_t H, K, X, b_size;
    Arr x, w, z_a, z_b, mask_poly, t_rows;
    MarlinStar star[3];
    Arr index_polys[12];
    CommitmentP index_cmts[12];
};
inline size_t marlin_max_degree(size_t n) { return 3 * next_pow2(n) + 8; }
inline std::vector<size_t> marlin_commit_sizes(size_t n) {
    const size_t H = next_pow2(n), K = H;
    return {H - 1, H + 1, 3 * H, H, H - 1, 2 * H, K - 1, 3 * K - 3, 3 * H - 1, H - 2, 3 * K - 4, K - 2, K};
}
inline MarlinInputs marlin_inputs(Machine& B, size_t n_constraints, uint64_t seed = 0x3A21) {
    MarlinInputs inp;
    const size_t H = next_pow2(n_constraints), K = H, X = 2, b_size = next_pow2(3 * K - 3);
    inp.H = H, inp.K = K, inp.X = X, inp.b_size = b_size;
    inp.x = B.random(seed + 4, X);
    inp.w = B.shared_copy(B.concat({B.random(seed + 1, H - X), B.zeros(1, X)}));
    inp.z_a = B.shared_copy(B.random(seed + 2, H));
    inp.z_b = B.shared_copy(B.random(seed + 3, H));
    inp.mask_poly = B.shared_copy(B.random(seed + 5, 3 * H));            // degree 3|H| + 2 zk - 3 with zk_bound = 1 (:376-380)
    inp.t_rows = B.random(seed + 6, H);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) inp.star[i].on_K[j] = B.random(seed + 10 * (i + 1) + j, K);
        for (int j = 0; j < 4; j++) inp.star[i].on_B[j] = B.random(seed + 10 * (i + 1) + 3 + j, b_size);
    }
    for (int j = 0; j < 12; j++) inp.index_polys[j] = B.ntt(B.random(seed + 100 + j, K), K, IFFT);   // row / col / val / row_col of A, B, C
    for (int j = 0; j < 12; j++) inp.index_cmts[j] = B.commit(inp.index_polys[j]);                   // committed at index time (the index_vk)
    B.transcript_point();
    return inp;
}

inline Arr mul_by_vanishing(Machine& B, const Arr& a, size_t n) {   // a (X^n - 1): a shifted up by n, minus a
    return B.sub(B.concat({B.zeros(a.lanes, n), a}), B.resized(a, a.n + n));
}

// Local compute of the three AHP prover rounds (marlin/src/ahp/prover.rs:300-704) and of Marlin::prove's commitments and openings
// (marlin/src/lib.rs:176-318).  Witness-side polynomials are share lanes, the arithmetised matrices and the third round are public.
inline Output marlin_prove(Machine& B, const MarlinInputs& inp) {
    const size_t H = inp.H, K = inp.K, X = inp.X, b_size = inp.b_size;
    Output out;
    std::map<std::string, Arr> blind;   // label -> blinding polynomial of a hiding commitment (share lanes, three coefficients)
    auto label_seed = [](const std::string& l) {
        uint64_t s = 0;
        for (unsigned char c : l) s += c;
        return (uint64_t)0xB11D + s;
    };
    // marlin_pc::commit -> KZG10::commit (kzg10/mod.rs:141-192); a hiding bound of Some(1) samples a blinding polynomial of degree 2, commits it over
    // powers_of_gamma_g (:181-186) and adds the two commitments (:188).  enqueue() starts the MSMs, publish() lists the commitment in the reference's order.
    auto enqueue = [&](const std::string& label, const Arr& a, bool hiding) {
        CommitmentP c = B.commit(a);
        if (hiding) {
            blind[label] = B.shared_copy(B.random(label_seed(label), 3));
            c = B.commit_sum(c, blind[label]);
        }
        return c;
    };
    auto publish = [&](const std::string& label, const CommitmentP& c) { out.commitments.emplace_back(label + "_cmt", c); };
    auto commit = [&](const std::string& label, const Arr& a, bool hiding) { publish(label, enqueue(label, a, hiding)); };
    // a + rand v_H: the zk blinding of the first-round polynomials (prover.rs:359-374), the blinding scalar a fixed constant
    auto mask = [&](const Arr& a, const char* tag, size_t n_dom) {
        const Fr rr = challenge(std::string("marlin.blind.") + tag);
        Arr bump = B.concat({B.constant(fr_neg(rr), 1), B.zeros(1, n_dom - 1), B.constant(rr, 1)});   // rr X^n - rr
        return B.plus(B.resized(a, n_dom + 1), bump);
    };
    // Schedule (see plonk_prove): statements and the order of `out` are the reference's; a statement is ENQUEUED as soon as the challenges it depends on are
    // known, and a transcript point waits for what the transcript has absorbed (mark / settle).
    // ---- first round (prover.rs:300-398) --------------------------------------------------------------------------------------------
    const Arr& mask_poly = inp.mask_poly;
    CommitmentP mask_cmt = enqueue("mask_poly", mask_poly, false);      // an input: its MSM runs while the round's other polynomials are interpolated
    Arr x_poly = B.ntt(inp.x, X, IFFT);                                  // public input polynomial (:324-330)
    Arr x_evals = B.ntt(x_poly, H, FFT);
    Arr w_poly = B.ntt(B.minus(inp.w, x_evals), H, IFFT);               // witness minus x on H, interpolated (:343-356)
    w_poly = B.div_vanishing(mask(w_poly, "w", H), X).first;            // / v_X (:357)
    commit("w", w_poly, true);                                           // hiding bounds Some(1), Some(1), Some(1), None (:386-390)
    Arr z_a = mask(B.ntt(inp.z_a, H, IFFT), "za", H);
    commit("z_a", z_a, true);
    Arr z_b = mask(B.ntt(inp.z_b, H, IFFT), "zb", H);
    commit("z_b", z_b, true);
    publish("mask_poly", mask_cmt);
    MarkP tp = B.mark();                                                 // transcript: the first round's four commitments
    // ---- second round (:439-556); what depends on no challenge comes first -------------------------------------------------------------
    Arr hpow = B.powers(B.root_of_unity(H), H);
    x_poly = B.ntt(inp.x, X, IFFT);                                      // interpolated again in the second round (:503-507)
    Arr z_poly = B.padded_add(mul_by_vanishing(B, w_poly, X), x_poly);  // w v_X + x (:512-517)
    const size_t z_c_n = z_a.n + z_b.n - 1;                              // z_c = z_a z_b, shared x shared (:466)
    const size_t summed_n = std::max(z_c_n, std::max(z_a.n, z_b.n));
    const size_t n_rhs = std::max(H + summed_n, H + z_poly.n) - 1;      // r_alpha_poly and t_poly have |H| coefficients
    const size_t mul_size = next_pow2(std::max(mask_poly.n, n_rhs + 1));   // GeneralEvaluationDomain::new(max(..)) (:522-531)
    auto ev = [&](const Arr& a) { return B.ntt(a, mul_size, FFT); };
    Arr z_poly_ev = ev(z_poly);
    // The reference forms z_c = z_a z_b, then summed = eta_c z_c + eta_a z_a + eta_b z_b (:468-476), and evaluates `summed` over the product domain (:526).
    // The transform is linear and multiplicative on exact field elements (deg z_c < |domain|), so those evaluations are
    // eta_c ev(z_a) ev(z_b) + eta_a ev(z_a) + eta_b ev(z_b): the two transforms need no challenge and run ahead of it; z_c's coefficients are used nowhere else
    Arr za_ev = ev(z_a), zb_ev = ev(z_b);
    Arr zc_ev = B.mul(za_ev, zb_ev);
    B.settle(tp);
    const Fr alpha = challenge("marlin.alpha"), eta_a = challenge("marlin.eta_a"), eta_b = challenge("marlin.eta_b"), eta_c = challenge("marlin.eta_c");
    // r(alpha, X) on H, unnormalised bivariate Lagrange: (alpha^|H| - 1) / (alpha - h^i)  (:480-482), public
    Arr r_alpha_evals = B.scale(B.inverse(B.add_const(B.scale(hpow, fr_neg(fr_one())), alpha)), vanishing(H, alpha));
    Arr t_poly = B.ntt(B.mul(inp.t_rows, r_alpha_evals), H, IFFT);      // calculate_t (:400-416)
    commit("t", t_poly, false);                                          // hiding bounds None, Some(1), None (:558-560)
    Arr r_alpha_poly = B.ntt(r_alpha_evals, H, IFFT);
    Arr summed_ev = B.lincomb({{eta_c, zc_ev}, {eta_a, za_ev}, {eta_b, zb_ev}});
    if (std::max(r_alpha_poly.n + summed_n, t_poly.n + z_poly.n) - 1 != n_rhs) throw czk::Panic(CZK_ERR_ARG, "pvm: marlin second-round sizes");
    Arr rhs = B.resized(B.ntt(B.sub(B.mul(ev(r_alpha_poly), summed_ev), B.mul(z_poly_ev, ev(t_poly))), mul_size, IFFT), n_rhs);
    Arr q_1 = B.padded_add(mask_poly, rhs);
    auto hx = B.div_vanishing(q_1, H);
    Arr h_1 = hx.first, g_1 = B.drop_first(hx.second, 1);
    hpow = z_poly_ev = summed_ev = za_ev = zb_ev = zc_ev = rhs = q_1 = r_alpha_evals = Arr();
    commit("g_1", g_1, true);
    commit("g_1_shifted", g_1, true);                                    // the degree bound's second commitment over the shifted powers, own blinding (marlin_pc/mod.rs:218-232)
    commit("h_1", h_1, false);
    // ---- third round (:585-704): everything public -----------------------------------------------------------------------------------
    B.transcript_point();
    const Fr beta = challenge("marlin.beta");
    // the witness of g_1's degree-bound opening at beta and the witness of its shifted randomness (marlin_pc/mod.rs:294-299 -> kzg10/mod.rs:200-224; used by
    // batch_open below) depend on beta alone: their MSM runs while the third round's public polynomials are computed
    Opening sh_beta = B.open_begin(g_1, beta), sh_rand_beta;
    const bool has_rand_beta = blind.count("g_1_shifted") != 0;
    if (has_rand_beta) sh_rand_beta = B.open_begin(blind["g_1_shifted"], beta);
    sh_beta = B.open_finish(std::move(sh_beta));
    if (has_rand_beta) sh_beta.proof = B.commit_sum(sh_beta.proof, sh_rand_beta.wit);
    const Fr vh = fr_mul(vanishing(H, alpha), vanishing(H, beta));
    const Fr etas[3] = {eta_a, eta_b, eta_c};
    const Fr minus_one = fr_neg(fr_one());
    // (each affine combination below is one pass -- Machine::lincomb with a constant -- where the reference's statement is a chain of vector operations;
    // the scalars eta_M and v_H(alpha) v_H(beta) multiply into one coefficient: the field is exact and associative)
    Arr den_b[3], val_b[3], val_inv[3];
    const Fr alpha_beta = fr_mul(beta, alpha);
    for (int i = 0; i < 3; i++) {
        const Arr &row = inp.star[i].on_K[0], &col = inp.star[i].on_K[1], &val = inp.star[i].on_K[2];
        Arr inv = B.inverse(B.mul(B.lincomb({{minus_one, row}}, &beta), B.lincomb({{minus_one, col}}, &alpha)));   // 1 / ((beta - row)(alpha - col))  (:612-620)
        val_inv[i] = B.mul(val, inv);
        const Arr &rb = inp.star[i].on_B[0], &cb = inp.star[i].on_B[1], &rcb = inp.star[i].on_B[2], &vb = inp.star[i].on_B[3];
        // beta alpha - r alpha - beta c + r_c  (:641-658)
        den_b[i] = B.lincomb({{fr_one(), rcb}, {fr_neg(alpha), rb}, {fr_neg(beta), cb}}, &alpha_beta);
        val_b[i] = vb;
    }
    Arr f = B.ntt(B.lincomb({{fr_mul(etas[0], vh), val_inv[0]}, {fr_mul(etas[1], vh), val_inv[1]}, {fr_mul(etas[2], vh), val_inv[2]}}), K, IFFT);
    Arr g_2 = B.drop_first(f, 1);
    commit("g_2", g_2, false);
    commit("g_2_shifted", g_2, false);                                   // degree bound |K| - 2
    const int others[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    std::vector<std::pair<Fr, Arr>> a_terms;
    for (int m = 0; m < 3; m++) a_terms.emplace_back(fr_mul(etas[m], vh), B.mul(val_b[m], B.mul(den_b[others[m][0]], den_b[others[m][1]])));   // (:664-673)
    Arr a_poly = B.resized(B.ntt(B.lincomb(a_terms), b_size, IFFT), 3 * K - 2);   // degree 3 |K| - 3 in a real index: 3 |K| - 2 coefficients
    Arr b_poly = B.resized(B.ntt(B.mul(den_b[0], B.mul(den_b[1], den_b[2])), b_size, IFFT), 3 * K - 2);
    Arr bf = B.poly_mul(b_poly, f);
    Arr h_2 = B.div_vanishing(B.sub(B.resized(a_poly, bf.n), bf), K).first;   // (a - b f) / v_K (:693-696)
    commit("h_2", h_2, false);
    // ---- evaluations and openings (marlin/src/lib.rs:262-318) -------------------------------------------------------------------------
    tp = B.mark();                                                       // transcript: the third round's commitments
    std::map<std::string, Arr> polys = {{"w", w_poly}, {"z_a", z_a}, {"z_b", z_b}, {"mask_poly", mask_poly}, {"t", t_poly},
                                        {"g_1", g_1},  {"h_1", h_1}, {"g_2", g_2}, {"h_2", h_2}};
    const char* mats[3] = {"a", "b", "c"};
    const char* parts[4] = {"row", "col", "val", "row_col"};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            const std::string name = std::string(mats[i]) + "_" + parts[j];
            polys[name] = inp.index_polys[4 * i + j];
            out.commitments.emplace_back(name + "_cmt", inp.index_cmts[4 * i + j]);   // index-time commitments (not prover work)
        }
    // the linear combinations of AHPForR1CS::construct_linear_combinations (marlin/src/ahp/mod.rs:115-260), coefficients = fixed stand-ins
    typedef std::vector<std::pair<Fr, std::string>> Terms;
    std::map<std::string, Terms> lcs;
    const Fr one = fr_one();
    lcs["z_b"] = {{one, "z_b"}};
    lcs["g_1"] = {{one, "g_1"}};
    lcs["t"] = {{one, "t"}};
    lcs["g_2"] = {{one, "g_2"}};
    lcs["outer_sumcheck"] = {{one, "mask_poly"}, {challenge("marlin.lc.z_a"), "z_a"}, {challenge("marlin.lc.w"), "w"}, {challenge("marlin.lc.h_1"), "h_1"}};
    lcs["inner_sumcheck"] = {{challenge("marlin.lc.a_val"), "a_val"}, {challenge("marlin.lc.b_val"), "b_val"}, {challenge("marlin.lc.c_val"), "c_val"},
                             {challenge("marlin.lc.h_2"), "h_2"}};
    for (int i = 0; i < 3; i++) {
        const std::string m = mats[i];
        lcs[m + "_denom"] = {{fr_neg(alpha), m + "_row"}, {fr_neg(beta), m + "_col"}, {one, m + "_row_col"}};
    }
    std::map<std::string, Fr> point = {{"beta", beta}};
    // verifier_query_set (ahp/verifier.rs:143-146, 207-211), labels in BTreeSet order
    std::map<std::string, std::vector<std::string>> query = {{"beta", {"g_1", "outer_sumcheck", "t", "z_b"}},
                                                             {"gamma", {"a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"}}};
    std::vector<ValueP> evals_beta, evals_gamma;
    // EvaluationsProvider::get_lc_eval (ahp/mod.rs:288-312): every polynomial of the combination is evaluated at the point.  construct_linear_combinations
    // evaluates what the coefficients of the two sumcheck combinations need (:155-157, :228-231), and Marlin::prove evaluates every queried combination
    // (lib.rs:283-292; the query set iterates in label order).  The evaluations at beta are enqueued here, ahead of gamma; those at gamma once it is drawn --
    // each list keeps the reference's order.
    auto evaluate_at = [&](const std::string& tag) {
        std::vector<ValueP>& into = tag == "beta" ? evals_beta : evals_gamma;
        auto lc_eval = [&](const std::string& label) {
            for (auto& t : lcs[label]) into.push_back(B.evaluate(polys[t.second], point[tag]));
        };
        for (auto& lt : std::vector<std::pair<const char*, const char*>>{{"z_b", "beta"}, {"t", "beta"}, {"g_1", "beta"}, {"a_denom", "gamma"}, {"b_denom", "gamma"},
                                                                         {"c_denom", "gamma"}, {"g_2", "gamma"}})
            if (tag == lt.second) lc_eval(lt.first);
        for (auto& kv : lcs) {   // std::map iterates in sorted label order, like Python's sorted(lcs)
            const bool in_beta = std::find(query["beta"].begin(), query["beta"].end(), kv.first) != query["beta"].end();
            if ((tag == "beta") == in_beta) lc_eval(kv.first);
        }
    };
    evaluate_at("beta");
    B.settle(tp);
    const Fr gamma = challenge("marlin.gamma");
    point["gamma"] = gamma;
    evaluate_at("gamma");
    tp = B.mark();                                                       // fs_rng.absorb(&evaluations) (:299)
    // what depends on gamma but not on the opening challenge: the witness of g_2's degree-bound opening at gamma ...
    std::map<std::string, Opening> sh_open = {{"beta", std::move(sh_beta)}}, sh_rand_open;
    if (has_rand_beta) sh_rand_open["beta"] = std::move(sh_rand_beta);
    {
        Opening sh = B.open_begin(g_2, gamma), sh_rand;
        const bool has_rand = blind.count("g_2_shifted") != 0;
        if (has_rand) sh_rand = B.open_begin(blind["g_2_shifted"], gamma);
        sh = B.open_finish(std::move(sh));
        if (has_rand) {
            sh.proof = B.commit_sum(sh.proof, sh_rand.wit);
            sh_rand_open["gamma"] = std::move(sh_rand);
        }
        sh_open["gamma"] = std::move(sh);
    }
    B.settle(tp);
    out.evals.emplace_back("evals_beta", evals_beta);
    out.evals.emplace_back("evals_gamma", evals_gamma);
    const Fr ch = challenge("marlin.opening_challenge");
    // PC::open_combinations (poly-commit/src/marlin/mod.rs:213-300) forms one polynomial per combination (`poly += (*coeff, cur_poly.polynomial())`, :275),
    // then batch_open (poly-commit/src/lib.rs:597-640): per query point the queried polynomials are folded with powers of the opening challenge
    // and opened once (marlin_pc/mod.rs:259-316): sum_j ch^j sum_t c_t p_t = sum over `terms` of (ch^j c_t) p_t -- ONE pass over the operands per
    // query point (Machine::lincomb) instead of a materialised polynomial per combination; a degree-bounded polynomial takes two challenges and also opens its own witness polynomial over
    // the shifted powers (:291-310, :318-330: the witness AND the witness of its shifted randomness, marlin_pc/mod.rs:294-299 -> kzg10/mod.rs:200-224 --
    // enqueued above, as soon as their point was known)
    for (const char* tagc : {"beta", "gamma"}) {
        const std::string tag = tagc;
        Fr c = one;
        Terms terms;
        std::string shifted;
        for (const std::string& label : query[tag]) {
            for (auto& t : lcs[label]) terms.emplace_back(fr_mul(c, t.first), t.second);
            c = fr_mul(c, ch);
            if (label == "g_1" || label == "g_2") {
                shifted = label;
                c = fr_mul(c, ch);
            }
        }
        const bool has_rand = sh_rand_open.count(tag) != 0;
        std::vector<std::pair<Fr, Arr>> fold_terms, rand_terms;
        for (auto& t : terms) {
            fold_terms.emplace_back(t.first, polys[t.second]);
            auto it = blind.find(t.second);
            if (it != blind.end()) rand_terms.emplace_back(t.first, it->second);
        }
        Arr folded = B.lincomb(fold_terms);
        // the folded polynomial with the folded randomness (`r += (challenge_j, &rand.rand)`, :288; KZG10::open :313)
        Arr r_fold;
        if (!rand_terms.empty()) r_fold = B.lincomb(rand_terms);
        Opening o = B.open_begin(folded, point[tag]);
        Opening rw;
        if (r_fold) rw = B.open_begin(r_fold, point[tag]);
        o = B.open_finish(std::move(o));
        o.terms = terms;
        if (r_fold) {
            o.random_v = rw.value;
            o.proof = B.commit_sum(o.proof, rw.wit);
        }
        out.openings.emplace_back("open_" + tag, std::move(o));
        // ... and the shifted witness over the shifted powers, with the shifted randomness' witness (open_with_witness_polynomial, :318-330)
        Opening so = std::move(sh_open[tag]);
        so.has_of = true;
        so.of = has_rand ? shifted + "_shifted" : shifted;
        if (has_rand) so.random_v = sh_rand_open[tag].value;
        out.openings.emplace_back("open_" + tag + "_shifted", std::move(so));
    }
    B.transcript_point();
    out.opened = std::move(B.opened);
    B.opened.clear();
    return out;
}

// ---- a canonical text form of an Output (tests compare it with the Python host's, element by element) ---------------------------------
inline std::string hex_bytes(const void* p, size_t n) {
    static const char* d = "0123456789abcdef";
    const uint8_t* b = (const uint8_t*)p;
    std::string s;
    for (size_t i = 0; i < n; i++) s += d[b[i] >> 4], s += d[b[i] & 15];
    return s;
}
inline std::string dump_commitment(const Commitment& c) {
    std::string s = "[";
    for (size_t ln = 0; ln < c.inf.size(); ln++) s += std::string(ln ? ", " : "") + "\"" + hex_bytes(&c.aff[12 * ln], 96) + (c.inf[ln] ? "01" : "00") + "\"";
    return s + "]";
}
inline std::string dump_value(const Value& v) {
    std::string s = "[";
    for (size_t ln = 0; ln < v.v.size(); ln++) s += std::string(ln ? ", " : "") + "\"" + hex_bytes(v.v[ln].l, 32) + "\"";
    return s + "]";
}
inline std::string dump_output(const Output& out) {
    std::string s = "{";
    bool first = true;
    auto key = [&](const std::string& k) {
        s += std::string(first ? "" : ", ") + "\"" + k + "\": ";
        first = false;
    };
    if (!out.opened.empty()) {
        key("opened");
        s += "[";
        for (size_t i = 0; i < out.opened.size(); i++) {
            Value v;
            v.v = out.opened[i];
            s += (i ? ", " : "") + dump_value(v);
        }
        s += "]";
    }
    for (auto& kv : out.commitments) key(kv.first), s += dump_commitment(*kv.second);
    for (auto& kv : out.evals) {
        key(kv.first);
        s += "[";
        for (size_t i = 0; i < kv.second.size(); i++) s += (i ? ", " : "") + dump_value(*kv.second[i]);
        s += "]";
    }
    for (auto& kv : out.openings) {
        const Opening& o = kv.second;
        key(kv.first);
        s += "{\"value\": " + dump_value(*o.value) + ", \"point\": \"" + hex_bytes(o.point.l, 32) + "\", \"proof\": " + dump_commitment(*o.proof);
        s += std::string(", \"of\": ") + (o.has_of ? "\"" + o.of + "\"" : "null");
        if (o.random_v) s += ", \"random_v\": " + dump_value(*o.random_v);
        if (!o.terms.empty()) {
            s += ", \"terms\": [";
            for (size_t i = 0; i < o.terms.size(); i++) s += std::string(i ? ", " : "") + "[\"" + hex_bytes(o.terms[i].first.l, 32) + "\", \"" + o.terms[i].second + "\"]";
            s += "]";
        }
        s += "}";
    }
    return s + "}";
}

}  // namespace pvm

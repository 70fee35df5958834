// groth16_host.hpp -- a torch-free, Python-free host of the Groth16 per-party local compute (BASELINE configs[1]) written
// against include/czk.hpp only: what the reference's Rust prover does once it links the shim (rust/czk), here in C++.
//
//   reference                                                                      here
//   R1CStoQAP::witness_map            mpc-snarks/src/groth/r1cs_to_qap.rs:47-113   Groth16Host::step (czk::R1CStoQAP::witness_map on DeviceLanes)
//   F::batch_product_in_place -> S::batch_mul (Beaver)   share/field.rs:97-127     Groth16Host::batch_product
//   SpdzFieldShare::batch_open                           share/spdz.rs:166-185     Groth16Host::open: all parties' lanes on this GPU -> lane sums;
//                                                                                  one party per process -> czk::Net::spdz_batch_open (czk_net)
//   create_proof's five MSMs          mpc-snarks/src/groth/prover.rs:104-156       czk::multi_scalar_mul_async
//   create_proof's O(1) group steps, calculate_coeff, Proof{a, b, c}  prover.rs:110-178, 216-232   Groth16Host::create_proof (public r, s)
//
// Inputs are host `std::vector`s, as the reference holds `Vec`s: the share lanes of the assignment go up ONCE
// (DeviceLanes::upload), every transform / pointwise step / MSM runs on the resident lanes, the 20 group elements of a proof come
// down.  The synthetic circuit, key and shares are the ones collaborative-zksnark_amd/provers.py::Groth16Local builds from the same
// seeds (SplitMix64 streams), so the two hosts must produce the same proof elements.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>

#include "czk.hpp"

namespace g16 {

using czk::Fr;

// ---- the caller's own field library (ark-ff in the reference): just enough host arithmetic to build inputs -----------------
namespace hostfr {
typedef unsigned __int128 u128;
static const uint64_t R_MOD[4] = {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};   // fr.rs:33-38
inline uint64_t inv64() {   // -r^-1 mod 2^64
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - R_MOD[0] * x;
    return (uint64_t)0 - x;
}
inline bool geq_r(const uint64_t* a) {
    for (int i = 3; i >= 0; i--)
        if (a[i] != R_MOD[i]) return a[i] > R_MOD[i];
    return true;
}
inline void sub_r(uint64_t* a) {
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - R_MOD[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
inline Fr add(const Fr& a, const Fr& b) {
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq_r(r.l)) sub_r(r.l);   // r < 2^253: no carry out of the top limb
    return r;
}
inline Fr sub(const Fr& a, const Fr& b) {
    Fr r;
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.l[i] + R_MOD[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
inline Fr mont_mul(const Fr& a, const Fr& b) {   // a b 2^-256 mod r (CIOS)
    static const uint64_t INV = inv64();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((u128)m * R_MOD[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * R_MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_r(r.l)) sub_r(r.l);
    return r;
}
inline Fr r2() {   // 2^512 mod r
    Fr x{{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) x = add(x, x);
    return x;
}
inline Fr from_repr(const Fr& canonical) {
    static const Fr R2 = r2();
    return mont_mul(canonical, R2);
}
inline Fr one() { return from_repr(Fr{{1, 0, 0, 0}}); }
}  // namespace hostfr

// SplitMix64 stream -> canonical values < r: top 3 bits masked (REPR_SHAVE_BITS, fr.rs:44), candidates >= r rejected
// (fields/arithmetic.rs:199-214).  Same stream as provers.py / tests/util.py rand_fr_canonical (chunk sizes included).
inline std::vector<Fr> rand_fr_canonical(uint64_t seed, size_t n) {
    std::vector<Fr> out;
    out.reserve(n);
    for (uint64_t chunk = 0; out.size() < n; chunk++) {
        size_t m = (size_t)((double)(n - out.size()) * 1.7) + 8;
        if (m < 16) m = 16;
        const uint64_t s0 = seed + 0x1000003ull * chunk;
        for (size_t i = 0; i < m; i++) {
            Fr v;
            for (int j = 0; j < 4; j++) {
                uint64_t z = s0 + (uint64_t)(4 * i + j + 1) * 0x9E3779B97F4A7C15ull;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                v.l[j] = z ^ (z >> 31);
            }
            v.l[3] &= ~(uint64_t)0 >> 3;
            if (!hostfr::geq_r(v.l)) out.push_back(v);
        }
    }
    out.resize(n);
    return out;
}

struct ProofElements {   // the group elements of one proof, per share lane (Jacobian, as the MSM returns them)
    std::vector<czk::G1Projective> h, l, a, b_g1;
    std::vector<czk::G2Projective> b_g2;
    explicit ProofElements(size_t lanes) : h(lanes), l(lanes), a(lanes), b_g1(lanes), b_g2(lanes) {}
};

struct ProofShare {   // one share lane's part of Proof{a, b, c} (groth16/src/data_structures.rs:13-20), Jacobian
    czk::G1Projective a;
    czk::G2Projective b;
    czk::G1Projective c;
};

// Discrete logs of a proving key to use in place of the synthetic one (canonical limbs): the queries from index 1 on (h: D - 1, l: N, a / b: N + 1; b serves
// b_g1 and b_g2), [alpha, beta, delta, a_query[0]] in G1 and [beta, delta] in G2 -- how tests/test_verify.py hands the compiled host a REAL key generated from known
// toxic waste (groth16/src/generator.rs:60-230), so that its proof can be put through the verification equation.
struct KeyScalars {
    std::vector<Fr> h, l, a, b, pk_g1, pk_g2;
};

class Groth16Host {
  public:
    static constexpr uint64_t BASE_SEED = 0xBA5E5;
    const czk::Context& ctx;
    size_t N, P, L, D;
    unsigned log_d = 0;
    double register_s = 0, setup_s = 0;

    // local_parties: the MPC parties whose share lanes live on this GPU (empty = all of them, BASELINE configs[1]).  With ONE local
    // party and a czk::Net this is the reference's own layout -- one process per party (mpc-net/src/multi.rs:15-23) -- and the two opens
    // of the witness map run SpdzFieldShare::batch_open's two broadcast rounds through the communicator, on lanes that stay in HBM.
    Groth16Host(const czk::Context& c, size_t n_constraints, size_t parties, uint64_t seed = 0xC0FFEE, bool no_tables = false,
                std::vector<size_t> local_parties = {}, const czk::Net* net = nullptr, bool commit_opens = true, const KeyScalars* key = nullptr)
        : ctx(c), N(n_constraints), P(parties), net_(net), commit_opens_(commit_opens) {
        auto t0 = std::chrono::steady_clock::now();
        if (local_parties.empty())
            for (size_t p = 0; p < P; p++) local_parties.push_back(p);
        local_ = local_parties;
        L = 2 * local_.size();
        if (local_.size() < P && (local_.size() != 1 || !net_ || net_->n_parties() != P || net_->party_id() != local_[0]))
            throw czk::Panic(CZK_ERR_ARG, "party layout: one local party per process, and a Net whose rank is that party");
        mac_share_ = local_[0] == 0 ? hostfr::one() : Fr{{0, 0, 0, 0}};   // mac_share() = 1 on the king, 0 elsewhere (share/spdz.rs:30-37)
        while (((size_t)1 << log_d) < N + 2) log_d++;   // D = next_pow2(N + num_instance)  (r1cs_to_qap.rs:63-65)
        D = (size_t)1 << log_d;
        domain_.emplace(*czk::Radix2EvaluationDomain::create(ctx, N + 2));
        // ---- synthetic proving key: P_i = [k_i] G; b_query[1] has no B entry -> infinity (groth16/src/generator.rs:156-163) ----
        h_query_ = mk_bases<CZK_G1>(D - 1, 1, false, no_tables, key ? &key->h : nullptr);
        l_query_ = mk_bases<CZK_G1>(N, 2, false, no_tables, key ? &key->l : nullptr);
        a_query_ = mk_bases<CZK_G1>(N + 1, 3, false, no_tables, key ? &key->a : nullptr);
        b_g1_query_ = mk_bases<CZK_G1>(N + 1, 4, true, no_tables, key ? &key->b : nullptr);
        b_g2_query_ = mk_bases<CZK_G2>(N + 1, 5, true, no_tables, key ? &key->b : nullptr);
        // the rest of the key (groth16/src/data_structures.rs:132-149), synthetic like the queries: G1 [vk.alpha_g1, beta_g1, delta_g1, a_query[0]],
        // G2 [vk.beta_g2, vk.delta_g2]; b_g1_query[0] and b_g2_query[0] are infinity in the real key (the constant-one variable has no B entry)
        {
            std::vector<Fr> k1 = key ? key->pk_g1 : rand_fr_canonical(BASE_SEED + 6, 4), k2 = key ? key->pk_g2 : rand_fr_canonical(BASE_SEED + 7, 2);
            if (k1.size() != 4 || k2.size() != 2) throw czk::Panic(CZK_ERR_ARG, "key scalars: pk_g1 holds 4 and pk_g2 2 scalars");
            pk_g1_.resize(4 * 12);
            pk_g2_.resize(2 * 24);
            ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, k1[0].l, 4, pk_g1_.data(), CZK_MEM_HOST));
            ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G2, k2[0].l, 2, pk_g2_.data(), CZK_MEM_HOST));
        }
        // NTT tables of the witness-map domain and the MSM workspaces of this key's largest G1 and G2 calls: at key load, not inside the first proof
        ctx.reserve(log_d, L, h_query_->raw(), D, L);
        ctx.reserve(0, 0, b_g2_query_->raw(), N + 1, L);
        ctx.sync();
        // ---- squaring-circuit witness (mpc-snarks/src/proof.rs:304-344) and its additive shares (share/spdz.rs:150-162) -------
        std::vector<Fr> w(N + 1);
        w[0] = hostfr::from_repr(rand_fr_canonical(seed, 1)[0]);
        for (size_t i = 0; i < N; i++) w[i + 1] = hostfr::mont_mul(w[i], w[i]);   // w_N = the public output
        std::vector<std::vector<Fr>> sh(P);
        std::vector<Fr> rest = w;
        for (size_t p = 0; p + 1 < P; p++) {
            sh[p] = rand_fr_canonical(seed + 17 * (p + 1), N + 1);
            for (size_t i = 0; i <= N; i++) {
                sh[p][i] = hostfr::from_repr(sh[p][i]);
                rest[i] = hostfr::sub(rest[i], sh[p][i]);
            }
        }
        sh[P - 1] = std::move(rest);
        const Fr one = hostfr::one();
        // share lanes 2 j + m: party j, m = 0 sh / 1 mac; mac lane = sh * mac(), mac() = 1 (spdz.rs:41-47)
        full_ = lanes(N + 2);   // [1, out | w_0 .. w_{N-1}] (r1cs_to_qap.rs:56-61); Public(1) lifted to the king's lanes
        wit_ = lanes(N);        // l-MSM scalars
        asg_ = lanes(N + 1);    // a / b MSM scalars: [out, witness]
        {
            std::vector<Fr> f(N + 2), g(N + 1);
            for (size_t k = 0; k < local_.size(); k++) {
                const size_t j = local_[k];
                f[0] = j == 0 ? one : Fr{{0, 0, 0, 0}};
                f[1] = sh[j][N];
                std::copy(sh[j].begin(), sh[j].begin() + N, f.begin() + 2);
                g[0] = sh[j][N];
                std::copy(sh[j].begin(), sh[j].begin() + N, g.begin() + 1);
                for (size_t m = 0; m < 2; m++) {     // the witness lanes go up once
                    full_->upload(2 * k + m, f);
                    wit_->upload(2 * k + m, 0, sh[j].data(), N);
                    asg_->upload(2 * k + m, g);
                }
            }
        }
        // ---- the circuit's matrices: a_i = b_i = w_i, c_i = w_{i+1} (c_{N-1} = out), A carries the instance copy rows (:79-83) ----
        {
            std::vector<uint64_t> rp(N + 3);
            for (size_t i = 0; i < N + 3; i++) rp[i] = i;
            std::vector<uint32_t> ca(N + 2), cb(N), cc(N);
            for (size_t i = 0; i < N; i++) ca[i] = cb[i] = (uint32_t)(2 + i);
            ca[N] = 0;
            ca[N + 1] = 1;
            for (size_t i = 0; i + 1 < N; i++) cc[i] = (uint32_t)(3 + i);
            cc[N - 1] = 1;
            std::vector<Fr> ones(N + 2, one);
            mat_a_.reset(new czk::ConstraintMatrix(ctx, rp.data(), ca.data(), ones.data(), N + 2, N + 2, N + 2));
            mat_b_.reset(new czk::ConstraintMatrix(ctx, rp.data(), cb.data(), ones.data(), N, N, N + 2));
            mat_c_.reset(new czk::ConstraintMatrix(ctx, rp.data(), cc.data(), ones.data(), N, N, N + 2));
        }
        // ---- dummy Beaver triples (wire/field.rs:41-60): king holds (1, 1, 1), everyone else (0, 0, 0) ----
        tx_ = lanes(D), ty_ = lanes(D), tz_ = lanes(D);
        for (auto* t : {tx_.get(), ty_.get(), tz_.get()})
            for (size_t ln = 0; ln < L; ln++)
                if (king_lane(ln)) ctx.check(czk_fr_powers(ctx.raw(), one.l, nullptr, D, t->data(ln), CZK_MEM_DEVICE));
        a_ = lanes(D), b_ = lanes(D), c_ = lanes(D), ab_ = lanes(D);
        sx_.reset(new czk::DeviceLanes(ctx, 1, D));
        oy_.reset(new czk::DeviceLanes(ctx, 1, D));
        chk_.reset(new czk::DeviceLanes(ctx, 2, D));
        ctx.sync();
        setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // One proof's local compute; only enqueues unless sync (consecutive proofs pipeline on the context's streams).  Returns the
    // proof's own output buffers, valid after the next ctx.sync().
    ProofElements& step(bool sync = true) {
        results.emplace_back(L);
        ProofElements& r = results.back();
        // create_proof MSMs that depend only on the witness (prover.rs:108, 132-156): they overlap the witness map below
        // (the order of the four: l, b_g2, a, b_g1 -- with the G2 kernel second the witness map below, which only makes progress beside THAT kernel, is done
        // closer to the moment `h` is needed: 14.51 / 14.54 against 14.42 / 14.29 proofs/s with b_g2 first, same box; EXPERIMENTS.md section 14)
        czk::multi_scalar_mul_async(*l_query_, *wit_, N, r.l.data(), true);
        czk::multi_scalar_mul_async(*b_g2_query_, *asg_, N + 1, r.b_g2.data(), true);
        czk::multi_scalar_mul_async(*a_query_, *asg_, N + 1, r.a.data(), true, true);         // (the same `assignment` as b_g2: one digit sort where the keys'
        czk::multi_scalar_mul_async(*b_g1_query_, *asg_, N + 1, r.b_g1.data(), true, true);   //  layouts and points at infinity agree -- CZK_MEM_SAME_SCALARS)
        // evaluate_constraint over the share lanes of the full assignment (r1cs_to_qap.rs:67-83, 95-100)
        mat_a_->evaluate(*full_, *a_);
        mat_b_->evaluate(*full_, *b_);
        mat_c_->evaluate(*full_, *c_);
        czk::R1CStoQAP::witness_map(*domain_, *a_, *b_, *c_, *ab_,
                                    [this](czk::DeviceLanes& x, czk::DeviceLanes& y, czk::DeviceLanes& xy) { batch_product(x, y, xy); });
        // the h MSM (prover.rs:104) consumes the witness map's output; not stable: the next proof overwrites `ab`
        czk::multi_scalar_mul_async(*h_query_, *ab_, D, r.h.data(), false);
        if (sync) ctx.sync();
        return r;
    }

    // The rest of create_proof after the five MSMs (prover.rs:110-178) for PUBLIC r, s (canonical limbs) -- every step is then linear in the
    // shares: calculate_coeff (:216-232: initial + query[0] + acc + vk_param) for A, B1, B2 and C = s A + r B1 - r s delta + l_acc + h_acc.
    // A public group element meets a share through `shift`: the king's lanes (sh, and mac under the stand-in key 1) add it, the others do not.
    // Returns every local lane's share of the proof; the parties' sh lanes sum to Proof{a, b, c}.
    std::vector<ProofShare> create_proof(const ProofElements& e, const czk::BigInteger256& r, const czk::BigInteger256& s) const {
        auto mul1 = [&](const czk::G1Projective& p, const czk::BigInteger256& k) { czk::G1Projective o; ctx.check(czk_jac_scalar_mul(ctx.raw(), CZK_G1, p.x.l, k.l, CZK_SCALAR_CANONICAL, o.x.l)); return o; };
        auto mul2 = [&](const czk::G2Projective& p, const czk::BigInteger256& k) { czk::G2Projective o; ctx.check(czk_jac_scalar_mul(ctx.raw(), CZK_G2, p.x.c0.l, k.l, CZK_SCALAR_CANONICAL, o.x.c0.l)); return o; };
        auto add1 = [&](const czk::G1Projective& p, const czk::G1Projective& q) { czk::G1Projective o; ctx.check(czk_jac_add(ctx.raw(), CZK_G1, p.x.l, q.x.l, o.x.l)); return o; };
        auto add2 = [&](const czk::G2Projective& p, const czk::G2Projective& q) { czk::G2Projective o; ctx.check(czk_jac_add(ctx.raw(), CZK_G2, p.x.c0.l, q.x.c0.l, o.x.c0.l)); return o; };
        auto mix1 = [&](const czk::G1Projective& p, const uint64_t* aff, bool inf) { czk::G1Projective o; ctx.check(czk_jac_add_mixed(ctx.raw(), CZK_G1, p.x.l, aff, inf, o.x.l)); return o; };
        auto mix2 = [&](const czk::G2Projective& p, const uint64_t* aff, bool inf) { czk::G2Projective o; ctx.check(czk_jac_add_mixed(ctx.raw(), CZK_G2, p.x.c0.l, aff, inf, o.x.c0.l)); return o; };
        const czk::G1Projective zero1{};   // z == 0: the identity (is_zero tests z alone)
        const czk::G2Projective zero2{};
        const czk::G1Projective delta_g1 = mix1(zero1, &pk_g1_[2 * 12], false);        // pk.delta_g1.into_projective()
        const czk::G2Projective delta_g2 = mix2(zero2, &pk_g2_[1 * 24], false);
        const czk::G1Projective r_s_delta_g1 = mul1(mul1(delta_g1, r), s), r_g1 = mul1(delta_g1, r), s_g1 = mul1(delta_g1, s);   // :113-117, :128, :144
        const czk::G2Projective s_g2 = mul2(delta_g2, s);                                                                            // :156
        czk::G1Projective neg_rs;
        ctx.check(czk_jac_neg(ctx.raw(), CZK_G1, r_s_delta_g1.x.l, neg_rs.x.l));
        const uint64_t inf_aff[24] = {0};
        std::vector<ProofShare> out(L);
        for (size_t ln = 0; ln < L; ln++) {
            const bool king = king_lane(ln);
            // calculate_coeff: res = initial; res.add_assign_mixed(&query[0]); res += &acc; res.add_assign_mixed(&vk_param)   (:224-229)
            const czk::G1Projective g_a = king ? mix1(add1(mix1(r_g1, &pk_g1_[3 * 12], false), e.a[ln]), &pk_g1_[0 * 12], false) : e.a[ln];      // a_query[0], vk.alpha_g1
            const czk::G1Projective g1_b = king ? mix1(add1(mix1(s_g1, inf_aff, true), e.b_g1[ln]), &pk_g1_[1 * 12], false) : e.b_g1[ln];       // b_g1_query[0] = infinity, beta_g1
            const czk::G2Projective g2_b = king ? mix2(add2(mix2(s_g2, inf_aff, true), e.b_g2[ln]), &pk_g2_[0 * 24], false) : e.b_g2[ln];       // b_g2_query[0] = infinity, vk.beta_g2
            czk::G1Projective g_c = add1(mul1(g_a, s), mul1(g1_b, r));          // s_g_a + r_g1_b (:138, :158, :165-166)
            if (king) g_c = add1(g_c, neg_rs);                                  // g_c -= &r_s_delta_g1
            g_c = add1(add1(g_c, e.l[ln]), e.h[ln]);                            // += l_aux_acc, += h_acc
            out[ln] = ProofShare{g_a, g2_b, g_c};
        }
        return out;
    }

    // number of non-zero entries of the two MAC-check vectors (share/spdz.rs:176-183: the reference asserts zero).  Party layout: the
    // check is part of czk::Net::spdz_batch_open, which panics on a failure; the vectors stay zero.
    uint64_t mac_check_failures() const {
        uint64_t bad = 0;
        ctx.check(czk_fr_lanes_sum(ctx.raw(), chk_->data(), 1, 2 * D, nullptr, &bad));
        return bad;
    }
    const czk::DeviceLanes& h_lanes() const { return *ab_; }
    std::deque<ProofElements> results;   // deque: element addresses stay valid while the MSMs that write them are in flight

    const std::vector<size_t>& local_parties() const { return local_; }

  private:
    bool king_lane(size_t ln) const { return local_[ln / 2] == 0; }
    std::unique_ptr<czk::DeviceLanes> lanes(size_t len) { return std::unique_ptr<czk::DeviceLanes>(new czk::DeviceLanes(ctx, L, len)); }

    template <int GROUP>
    std::unique_ptr<czk::Bases<GROUP>> mk_bases(size_t n, uint64_t sd, bool inf_first, bool no_tables, const std::vector<Fr>* given = nullptr) {
        if (given && given->size() != n) throw czk::Panic(CZK_ERR_ARG, "key scalars: a query has the wrong length");
        std::vector<Fr> k = given ? *given : rand_fr_canonical(BASE_SEED + sd, n);
        std::vector<uint64_t> pts((GROUP == CZK_G1 ? 12 : 24) * n);
        ctx.check(czk_fixed_base_points(ctx.raw(), GROUP, k[0].l, n, pts.data(), CZK_MEM_HOST));
        std::vector<uint8_t> inf(n, 0);
        if (inf_first) inf[0] = 1;
        auto t0 = std::chrono::steady_clock::now();
        std::unique_ptr<czk::Bases<GROUP>> b(new czk::Bases<GROUP>(ctx, pts.data(), inf.data(), n, no_tables ? CZK_MEM_NO_TABLES : 0));
        register_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return b;
    }

    // one open with every party's lanes on this GPU: value = sum of the sh lanes; MAC-check vector = mac_share * value - sum of
    // the mac lanes with mac_share = 1 on the king, 0 elsewhere (share/spdz.rs:31-37, 166-185)
    void open(const czk::DeviceLanes& shares, czk::DeviceLanes& out, uint64_t* chk) {
        const int M = CZK_MEM_DEVICE;
        if (local_.size() < P) {   // one party per process: Net::broadcast(&s_vals), dx_t, Net::atomic_broadcast(&dx_ts), assert (spdz.rs:166-185)
            net_->spdz_batch_open(shares, 0, mac_share_, D, out.data(), commit_opens_);
            return;
        }
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_ADD, shares.data(0), shares.data(2), out.data(), D, M));
        for (size_t p = 2; p < P; p++) ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_ADD, out.data(), shares.data(2 * p), out.data(), D, M));
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_SUB, out.data(), shares.data(1), chk, D, M));
        for (size_t p = 1; p < P; p++) ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_SUB, chk, shares.data(2 * p + 1), chk, D, M));
    }

    // F::batch_product_in_place -> S::batch_mul (share/field.rs:97-127): (s + x), (o + y), two opens, local combine
    void batch_product(czk::DeviceLanes& a, czk::DeviceLanes& b, czk::DeviceLanes& ab) {
        const int M = CZK_MEM_DEVICE;
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_ADD, a.data(), tx_->data(), a.data(), L * D, M));
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_ADD, b.data(), ty_->data(), b.data(), L * D, M));
        open(a, *sx_, chk_->data(0));
        open(b, *oy_, chk_->data(1));
        for (size_t ln = 0; ln < L; ln++)
            ctx.check(czk_fr_beaver_combine(ctx.raw(), tx_->data(ln), ty_->data(ln), tz_->data(ln), sx_->data(), oy_->data(), king_lane(ln) ? 1 : 0, ab.data(ln), D, M));
    }

    const czk::Net* net_ = nullptr;
    bool commit_opens_ = true;   // Net::atomic_broadcast(&dx_ts) as in the reference (spdz.rs:179); false = the measured opt-out
    std::vector<size_t> local_;
    Fr mac_share_{};
    std::vector<uint64_t> pk_g1_, pk_g2_;
    std::optional<czk::Radix2EvaluationDomain> domain_;
    std::unique_ptr<czk::G1Bases> h_query_, l_query_, a_query_, b_g1_query_;
    std::unique_ptr<czk::G2Bases> b_g2_query_;
    std::unique_ptr<czk::DeviceLanes> full_, wit_, asg_, tx_, ty_, tz_, a_, b_, c_, ab_, sx_, oy_, chk_;
    std::unique_ptr<czk::ConstraintMatrix> mat_a_, mat_b_, mat_c_;
};

}  // namespace g16

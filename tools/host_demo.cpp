// host_demo.cpp -- exercises include/czk.hpp (the C++ mirror of the reference's trait surface) end to end.
// Build: g++ -std=c++17 -Iinclude tools/host_demo.cpp -Lcollaborative-zksnark_amd -lczk_hip -Wl,-rpath,$PWD/collaborative-zksnark_amd -o tools/host_demo.bin
#include <cstdio>
#include <cstring>

#include "czk.hpp"

using namespace czk;

static std::vector<Fr> fr_from_u64(const Context& ctx, const std::vector<uint64_t>& v) {
    std::vector<Fr> r(v.size());
    for (size_t i = 0; i < v.size(); i++) r[i] = Fr{{v[i], 0, 0, 0}};
    ctx.check(czk_fr_from_repr(ctx.raw(), r[0].l, r[0].l, r.size(), CZK_MEM_HOST));
    return r;
}
static bool eq(const Fr& a, const Fr& b) { return memcmp(a.l, b.l, 32) == 0; }
#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    Context ctx(0);
    // EvaluationDomain::new -> None beyond 2^47 (radix2/mod.rs:61-63)
    REQUIRE(!Radix2EvaluationDomain::create(ctx, (size_t)1 << 48).has_value());
    auto dom = Radix2EvaluationDomain::create(ctx, 13);
    REQUIRE(dom && dom->size() == 16);

    // fft / ifft and coset round trips on a ragged vector (13 coefficients, resized to 16)
    std::vector<uint64_t> raw;
    for (uint64_t i = 0; i < 13; i++) raw.push_back(i * i + 7);
    std::vector<Fr> x = fr_from_u64(ctx, raw), y = x;
    dom->fft_in_place(y);
    REQUIRE(y.size() == 16);
    dom->ifft_in_place(y);
    for (size_t i = 0; i < 13; i++) REQUIRE(eq(x[i], y[i]));
    for (size_t i = 13; i < 16; i++) REQUIRE(eq(y[i], Fr{{0, 0, 0, 0}}));
    y = x;
    dom->coset_fft_in_place(y);
    dom->coset_ifft_in_place(y);
    for (size_t i = 0; i < 13; i++) REQUIRE(eq(x[i], y[i]));
    // too-long input panics like the reference's assert (radix2/mod.rs:100)
    bool threw = false;
    try {
        std::vector<Fr> z(17);
        dom->fft_in_place(z);
    } catch (const Panic& p) {
        threw = p.code == CZK_ERR_SIZE;
    }
    REQUIRE(threw);

    // share vector: the sh lane of FFT(MpcField) is FFT(sh values); Public entries are lifted on the king
    std::vector<MpcField> mv(13);
    for (size_t i = 0; i < 13; i++) {
        mv[i].shared = i != 12;
        mv[i].sh = x[i];
        mv[i].mac = x[(i + 1) % 13];
    }
    std::vector<Fr> plain = x;
    dom->fft_in_place(plain);
    dom->fft_in_place(mv);
    for (size_t i = 0; i < 16; i++) REQUIRE(mv[i].shared && eq(mv[i].sh, plain[i]));

    // MSM: P_i = [i] G, scalars all one  =>  [n(n+1)/2] G ; and the two SPDZ lanes agree (spdz.rs:441-442)
    const size_t n = 100;
    std::vector<uint64_t> k(4 * (n + 1), 0);
    for (size_t i = 0; i < n; i++) k[4 * i] = i + 1;
    k[4 * n] = n * (n + 1) / 2;
    std::vector<uint64_t> pts(12 * (n + 1));
    ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, k.data(), n + 1, pts.data(), CZK_MEM_HOST));
    G1Bases bases(ctx, pts.data(), nullptr, n);
    REQUIRE(bases.len() == n);
    std::vector<uint64_t> ones_raw(n, 1);
    std::vector<Fr> ones = fr_from_u64(ctx, ones_raw);
    G1Projective acc = G1Affine::multi_scalar_mul(bases, ones);
    uint64_t aff[12];
    uint8_t inf = 1;
    ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, acc.x.l, 1, aff, &inf));
    REQUIRE(!inf && memcmp(aff, &pts[12 * n], 96) == 0);
    std::vector<BigInteger256> ones_repr(n, BigInteger256{{1, 0, 0, 0}});
    G1Projective acc2 = VariableBaseMSM::multi_scalar_mul(bases, ones_repr);
    ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, acc2.x.l, 1, aff, &inf));
    REQUIRE(!inf && memcmp(aff, &pts[12 * n], 96) == 0);
    std::vector<MpcField> sc(n);
    for (size_t i = 0; i < n; i++) { sc[i].shared = true; sc[i].sh = ones[i]; sc[i].mac = x[i % 13]; }
    SpdzGroupShareG1 gs = SpdzGroupShareG1::multi_scale_pub_group(bases, sc);
    uint64_t a2[24];
    uint8_t i2[2];
    G1Projective both[2] = {gs.sh, gs.mac};
    ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, both[0].x.l, 2, a2, i2));
    REQUIRE(memcmp(a2, a2 + 12, 96) == 0 && memcmp(a2, &pts[12 * n], 96) == 0);

    // KZG10::commit: commitment to the all-ones polynomial with an all-ones blinding polynomial over the same powers
    // = 2 * [n(n+1)/2] G
    {
        G1Projective cm = KZG10::commit(bases, ones, &bases, &ones);
        std::vector<uint64_t> k2(4, 0);
        k2[0] = n * (n + 1);
        uint64_t want[12];
        ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, k2.data(), 1, want, CZK_MEM_HOST));
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, cm.x.l, 1, aff, &inf));
        REQUIRE(!inf && memcmp(aff, want, 96) == 0);
    }

    // witness map of the 6-constraint squaring circuit (proof.rs:304-344): quotient is exact => h[D-1] == 0
    const size_t N = 6;
    auto d8 = Radix2EvaluationDomain::create(ctx, N + 2);
    REQUIRE(d8 && d8->size() == 8);
    std::vector<Fr> w = fr_from_u64(ctx, {3});
    for (size_t i = 0; i < N; i++) {
        Fr sq;
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_MUL, w[i].l, w[i].l, sq.l, 1, CZK_MEM_HOST));
        w.push_back(sq);
    }
    std::vector<Fr> one = fr_from_u64(ctx, {1});
    std::vector<Fr> a(w.begin(), w.begin() + N), b = a, c(w.begin() + 1, w.begin() + N + 1);
    a.push_back(one[0]);
    a.push_back(w[N]);
    std::vector<Fr> h = R1CStoQAP::witness_map(ctx, *d8, a, b, c);
    REQUIRE(h.size() == 8 && eq(h[7], Fr{{0, 0, 0, 0}}));
    printf("host_demo OK\n");
    return 0;
}

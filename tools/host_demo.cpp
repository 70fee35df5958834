// host_demo.cpp -- exercises include/czk.hpp (the C++ mirror of the reference's trait surface) end to end.
// Build: g++ -std=c++17 -Iinclude tools/host_demo.cpp -Lcollaborative-zksnark_amd -lczk_hip -Wl,-rpath,$PWD/collaborative-zksnark_amd -o tools/host_demo.bin
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

#include <thread>

#include "czk.hpp"
#include "groth16_host.hpp"
#include "polyvm_host.hpp"

using namespace czk;

static std::vector<Fr> fr_from_u64(const Context& ctx, const std::vector<uint64_t>& v) {
    std::vector<Fr> r(v.size());
    for (size_t i = 0; i < v.size(); i++) r[i] = Fr{{v[i], 0, 0, 0}};
    ctx.check(czk_fr_from_repr(ctx.raw(), r[0].l, r[0].l, r.size(), CZK_MEM_HOST));
    return r;
}
static bool eq(const Fr& a, const Fr& b) { return memcmp(a.l, b.l, 32) == 0; }
#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

// `host_demo.bin dump-lift`: prints the share lanes of a mixed Public / Shared vector after coset_fft and after ifft, once as
// the king and once as another party, for tests/test_abi.py to compare with the CHECKER (oracle/) applied to the lifted lanes --
// the C++ mirror is checked against the reference restatement, not against itself.
static void print_fr(const char* tag, const Fr& f) { printf("%s %016lx %016lx %016lx %016lx\n", tag, (unsigned long)f.l[0], (unsigned long)f.l[1], (unsigned long)f.l[2], (unsigned long)f.l[3]); }
static int dump_lift(const Context& ctx) {
    std::vector<uint64_t> raw;
    for (uint64_t i = 0; i < 26; i++) raw.push_back(0x9e3779b97f4a7c15ull * (i + 1) >> 7);
    std::vector<Fr> x = fr_from_u64(ctx, raw);
    for (int king = 1; king >= 0; king--) {
        auto dom = Radix2EvaluationDomain::create(ctx, 13, king != 0);
        for (int kind = 0; kind < 2; kind++) {
            std::vector<MpcField> mv(13);
            for (size_t i = 0; i < 13; i++) {
                mv[i].shared = (i % 3) != 1;            // entries 1, 4, 7, 10 are Public
                mv[i].sh = x[i];
                mv[i].mac = x[13 + i];
            }
            if (kind == 0) dom->coset_fft_in_place(mv);
            else dom->ifft_in_place(mv);
            printf("case king=%d kind=%s\n", king, kind == 0 ? "coset_fft" : "ifft");
            for (size_t i = 0; i < 16; i++) {
                print_fr("sh", mv[i].sh);
                print_fr("mac", mv[i].mac);
            }
        }
    }
    return 0;
}

// --transport rccl | shm | ipc (include/czk.h czk_net_transport)
static int transport_of(const char* s) { return !strcmp(s, "rccl") ? CZK_NET_RCCL : !strcmp(s, "ipc") ? CZK_NET_IPC : CZK_NET_SHM; }
static const char* transport_name(int t) { return t == CZK_NET_RCCL ? "rccl" : t == CZK_NET_IPC ? "ipc" : "shm"; }
static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) s += d[p[i] >> 4], s += d[p[i] & 15];
    return s;
}
static std::vector<uint8_t> unhex(const char* h) {
    std::vector<uint8_t> v;
    auto nib = [](char c) { return (uint8_t)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10); };
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) v.push_back((uint8_t)(nib(h[i]) << 4 | nib(h[i + 1])));
    return v;
}
// The affine group elements of one proof on this process's lanes as bench.py's `results_sha256` orders them: per query (h, l, a, b_g1,
// b_g2) the lanes in order, each as its affine limbs followed by the infinity byte.  Returns the five per-query byte strings.
static std::vector<std::vector<uint8_t>> proof_bytes(const Context& ctx, const g16::ProofElements& r, size_t L) {
    std::vector<std::vector<uint8_t>> out(5);
    const G1Projective* g1[4] = {r.h.data(), r.l.data(), r.a.data(), r.b_g1.data()};
    for (int q = 0; q < 5; q++) {
        const size_t aw = q < 4 ? 12 : 24;
        std::vector<uint64_t> aff(L * aw);
        std::vector<uint8_t> inf(L);
        if (q < 4) ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, g1[q]->x.l, L, aff.data(), inf.data()));
        else ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G2, r.b_g2.data()->x.c0.l, L, aff.data(), inf.data()));
        for (size_t ln = 0; ln < L; ln++) {
            const uint8_t* b = (const uint8_t*)&aff[ln * aw];
            out[q].insert(out[q].end(), b, b + 8 * aw);
            out[q].push_back(inf[ln]);
        }
    }
    return out;
}

// `host_demo.bin bench [--log-n K | --constraints N] [--parties P] [--steps K] [--warmup W] [--no-tables] [--dump FILE]`:
// BASELINE configs[1] (Groth16, SPDZ, both parties' lanes on one GPU) driven from this process alone -- no torch, no Python in
// the timed loop: host vectors -> share lanes up once -> K pipelined proofs -> 20 group elements per proof down.  Prints one JSON
// line; bench.py reports it as `seam_device_handles`.  --dump writes the h lanes and the proof elements of the last proof for
// tests/test_gpu_parity.py to compare with the checker: u64 header {N, D, lanes}, lanes x D Fr, then per query (h, l, a, b_g1:
// lanes x 12 u64 affine + lanes flag bytes; b_g2: lanes x 24 u64 + flags), then per lane its share of Proof{a, b, c} for public r, s
// (Groth16Host::create_proof: a 12 u64 + flag, b 24 u64 + flag, c 12 u64 + flag).
static int bench(const Context& ctx, int argc, char** argv) {
    size_t n = (size_t)1 << 20, parties = 2, steps = 20, warmup = 2;
    bool no_tables = false;
    const char* dump = nullptr;
    const char* key_file = nullptr;
    for (int i = 2; i < argc; i++) {
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (!strcmp(argv[i], "--key-file")) key_file = val();
        else if (!strcmp(argv[i], "--log-n")) n = (size_t)1 << atoi(val());
        else if (!strcmp(argv[i], "--constraints")) n = (size_t)atoll(val());
        else if (!strcmp(argv[i], "--parties")) parties = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--steps")) steps = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--warmup")) warmup = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--no-tables")) no_tables = true;
        else if (!strcmp(argv[i], "--dump")) dump = val();
        else { printf("unknown argument %s\n", argv[i]); return 2; }
    }
    if (n < 2 || parties < 2 || steps < 1) { printf("bench: need --constraints >= 2, --parties >= 2, --steps >= 1\n"); return 2; }
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    // --key-file: the discrete logs of a REAL proving key (tests/test_verify.py): u64 header [N, D], then h (D - 1), l (N), a (N + 1), b (N + 1), pk_g1 (4),
    // pk_g2 (2) as canonical 4 x u64 scalars
    g16::KeyScalars key;
    if (key_file) {
        FILE* f = fopen(key_file, "rb");
        REQUIRE(f != nullptr);
        uint64_t hdr[2];
        REQUIRE(fread(hdr, 8, 2, f) == 2 && hdr[0] == n);
        const size_t Dk = (size_t)hdr[1];
        auto rd = [&](std::vector<g16::Fr>& v, size_t cnt) {
            v.resize(cnt);
            REQUIRE(fread(v.data(), 32, cnt, f) == cnt);
        };
        rd(key.h, Dk - 1), rd(key.l, n), rd(key.a, n + 1), rd(key.b, n + 1), rd(key.pk_g1, 4), rd(key.pk_g2, 2);
        fclose(f);
    }
    g16::Groth16Host prover(ctx, n, parties, 0xC0FFEE, no_tables, {}, nullptr, false, key_file ? &key : nullptr);
    auto t0 = clk::now();
    prover.step();                                            // first proof: also builds the NTT tables and sizes the workspaces
    const double first_ms = secs(t0, clk::now()) * 1e3;
    for (size_t i = 1; i < warmup; i++) prover.step();
    t0 = clk::now();
    prover.step();                                            // one proof alone: enqueue -> results on the host
    const double latency_ms = secs(t0, clk::now()) * 1e3;
    prover.results.clear();
    ctx.sync();
    t0 = clk::now();
    for (size_t i = 0; i < steps; i++) prover.step(false);    // consecutive proofs pipeline on the context's streams
    ctx.sync();
    const double dt = secs(t0, clk::now());
    // every pipelined proof has the same inputs: all must yield the same group elements (affine: Jacobian triples differ)
    const size_t L = prover.L;
    std::vector<uint64_t> ref_aff, aff;
    std::vector<uint8_t> ref_inf, inf;
    auto affine = [&](const g16::ProofElements& r, std::vector<uint64_t>& a, std::vector<uint8_t>& f) {
        a.assign(L * (4 * 12 + 24), 0);
        f.assign(L * 5, 0);
        const G1Projective* g1[4] = {r.h.data(), r.l.data(), r.a.data(), r.b_g1.data()};
        for (int q = 0; q < 4; q++) ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, g1[q]->x.l, L, &a[q * L * 12], &f[q * L]));
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G2, r.b_g2.data()->x.c0.l, L, &a[4 * L * 12], &f[4 * L]));
    };
    REQUIRE(prover.results.size() == steps);
    for (size_t k = 0; k < steps; k++) {
        affine(prover.results[k], aff, inf);
        if (k == 0) { ref_aff = aff; ref_inf = inf; }
        REQUIRE(aff == ref_aff && inf == ref_inf);
    }
    const uint64_t bad = prover.mac_check_failures();
    REQUIRE(bad == 0);
    uint8_t digest[32];
    {
        std::vector<uint8_t> all;
        for (const auto& q : proof_bytes(ctx, prover.results[0], L)) all.insert(all.end(), q.begin(), q.end());
        czk_sha256(all.data(), all.size(), digest);
    }
    if (dump) {
        FILE* f = fopen(dump, "wb");
        REQUIRE(f != nullptr);
        const uint64_t hdr[3] = {prover.N, prover.D, L};
        fwrite(hdr, 8, 3, f);
        std::vector<Fr> h(prover.D);
        for (size_t ln = 0; ln < L; ln++) {
            prover.h_lanes().download(ln, 0, h.data(), prover.D);
            fwrite(h.data(), 32, prover.D, f);
        }
        for (int q = 0; q < 5; q++) {
            const size_t aw = q < 4 ? 12 : 24;
            fwrite(&ref_aff[q * L * 12], 8, L * aw, f);
            fwrite(&ref_inf[q * L], 1, L, f);
        }
        // ... then every lane's share of Proof{a, b, c} for the public r, s = rand_fr_canonical(0xC0FFEE + 99, 2): a, c as 12 u64 + flag, b as 24 u64 + flag
        std::vector<Fr> rs = g16::rand_fr_canonical(0xC0FFEE + 99, 2);
        const BigInteger256 r{{rs[0].l[0], rs[0].l[1], rs[0].l[2], rs[0].l[3]}}, s_{{rs[1].l[0], rs[1].l[1], rs[1].l[2], rs[1].l[3]}};
        for (const g16::ProofShare& ps : prover.create_proof(prover.results[0], r, s_)) {
            uint64_t a1[12], a2[24];
            uint8_t fl = 0;
            ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, ps.a.x.l, 1, a1, &fl));
            fwrite(a1, 8, 12, f), fwrite(&fl, 1, 1, f);
            ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G2, ps.b.x.c0.l, 1, a2, &fl));
            fwrite(a2, 8, 24, f), fwrite(&fl, 1, 1, f);
            ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, ps.c.x.l, 1, a1, &fl));
            fwrite(a1, 8, 12, f), fwrite(&fl, 1, 1, f);
        }
        fclose(f);
    }
    printf("{\"harness\": \"tools/host_demo.cpp bench (C++ over include/czk.hpp; no torch, no Python)\", \"constraints\": %zu, \"parties\": %zu, "
           "\"share_lanes\": %zu, \"steps\": %zu, \"warmup\": %zu, \"ms_per_proof\": %.4f, \"proofs_per_s\": %.5f, \"latency_ms_single_proof\": %.3f, "
           "\"first_proof_ms\": %.3f, \"register_key_s\": %.4f, \"setup_s\": %.3f, \"window_tables\": %s, \"pipelined_proofs_equal\": true, "
           "\"mac_check_failures\": %llu, \"results_sha256\": \"%s\"}\n",
           prover.N, prover.P, L, steps, warmup, dt / steps * 1e3, steps / dt, latency_ms, first_ms, prover.register_s, prover.setup_s,
           no_tables ? "false" : "true", (unsigned long long)bad, hex(digest, 32).c_str());
    return 0;
}

// `host_demo.bin party --rank R --world W --id HEX [--transport shm|rccl] [--device D] [--log-n K | --constraints N] [--steps K]
// [--warmup W] [--no-tables] [--no-commit-opens] [--exchange ring|p2p]`: ONE MPC party of the reference's own layout -- one process per
// party (mpc-net/src/multi.rs:15-23) -- as a compiled host: party R's two share lanes on this process's GPU, the two opens of every
// witness map as SpdzFieldShare::batch_open through czk::Net (RCCL between GPUs; shared memory when the parties share a GPU), nothing
// leaves HBM except the proof's group elements.  Rank 0 prints one JSON line whose results_sha256 covers ALL parties' elements in
// bench.py's order (the other ranks' bytes reach it with send_bytes_to_king), so it equals the one-GPU layout's digest.
// `host_demo.bin party-launch --world W [...]` forks + execs the W ranks with a fresh id.
static int party(int argc, char** argv) {
    size_t n = (size_t)1 << 20, steps = 4, warmup = 1;
    int rank = -1, world = 0, device = -1, transport = CZK_NET_SHM, exchange = 0;
    bool no_tables = false, commit = true;   // dx_ts through atomic_broadcast, as the reference does (spdz.rs:179); --no-commit-opens is the reported opt-out
    std::vector<uint8_t> id;
    for (int i = 2; i < argc; i++) {
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (!strcmp(argv[i], "--log-n")) n = (size_t)1 << atoi(val());
        else if (!strcmp(argv[i], "--constraints")) n = (size_t)atoll(val());
        else if (!strcmp(argv[i], "--steps")) steps = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--warmup")) warmup = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--rank")) rank = atoi(val());
        else if (!strcmp(argv[i], "--world")) world = atoi(val());
        else if (!strcmp(argv[i], "--device")) device = atoi(val());
        else if (!strcmp(argv[i], "--id")) id = unhex(val());
        else if (!strcmp(argv[i], "--transport")) transport = transport_of(val());
        else if (!strcmp(argv[i], "--exchange")) exchange = !strcmp(val(), "p2p") ? 1 : 0;
        else if (!strcmp(argv[i], "--no-tables")) no_tables = true;
        else if (!strcmp(argv[i], "--commit-opens")) commit = true;
        else if (!strcmp(argv[i], "--no-commit-opens")) commit = false;
        else { printf("unknown argument %s\n", argv[i]); return 2; }
    }
    if (rank < 0 || world < 2 || rank >= world || id.empty() || n < 2 || steps < 1) { printf("party: need --rank, --world >= 2, --id\n"); return 2; }
    if (device < 0) device = transport == CZK_NET_RCCL ? rank : 0;
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    Context ctx(device);
    Net net(ctx, transport, rank, world, id);
    net.set_option("exchange", exchange);
    g16::Groth16Host prover(ctx, n, (size_t)world, 0xC0FFEE, no_tables, {(size_t)rank}, &net, commit);
    for (size_t i = 0; i < warmup; i++) prover.step();
    prover.results.clear();
    net.reset_stats();
    net.barrier();
    auto t0 = clk::now();
    for (size_t i = 0; i < steps; i++) prover.step(false);
    ctx.sync();
    net.barrier();
    const double dt = secs(t0, clk::now());
    const Net::Stats st = net.stats();
    const size_t L = prover.L;
    auto mine = proof_bytes(ctx, prover.results[0], L);
    for (size_t k = 1; k < steps; k++) REQUIRE(proof_bytes(ctx, prover.results[k], L) == mine);   // equal inputs, equal proofs
    std::vector<uint8_t> flat;
    for (const auto& q : mine) flat.insert(flat.end(), q.begin(), q.end());
    auto all = net.send_bytes_to_king(flat);
    if (rank != 0) return 0;
    std::vector<uint8_t> ordered;            // per query, the parties in order (bench.py's order)
    size_t off = 0;
    for (int q = 0; q < 5; q++) {
        const size_t len = mine[q].size();
        for (int p = 0; p < world; p++) ordered.insert(ordered.end(), (*all)[p].begin() + off, (*all)[p].begin() + off + len);
        off += len;
    }
    uint8_t digest[32];
    czk_sha256(ordered.data(), ordered.size(), digest);
    printf("{\"harness\": \"tools/host_demo.cpp party (C++ over include/czk.hpp: one process per MPC party, opens through czk_net)\", \"layout\": \"party\", "
           "\"transport\": \"%s\", \"exchange\": \"%s\", \"constraints\": %zu, \"parties\": %d, \"share_lanes_per_process\": %zu, \"steps\": %zu, "
           "\"warmup\": %zu, \"ms_per_proof\": %.4f, \"proofs_per_s\": %.5f, \"window_tables\": %s, \"commit_opens\": %s, \"pipelined_proofs_equal\": true, "
           "\"king_net_stats\": {\"bytes_sent\": %llu, \"bytes_recv\": %llu, \"broadcasts\": %llu, \"to_king\": %llu, \"from_king\": %llu}, "
           "\"results_sha256\": \"%s\"}\n",
           transport_name(transport), exchange ? "p2p" : "ring", prover.N, world, L, steps, warmup, dt / steps * 1e3, steps / dt,
           no_tables ? "false" : "true", commit ? "true" : "false", (unsigned long long)st.bytes_sent, (unsigned long long)st.bytes_recv,
           (unsigned long long)st.broadcasts, (unsigned long long)st.to_king, (unsigned long long)st.from_king, hex(digest, 32).c_str());
    return 0;
}

static int party_launch(int argc, char** argv) {
    int world = 0, transport = CZK_NET_SHM;
    for (int i = 2; i + 1 < argc; i++) {
        if (!strcmp(argv[i], "--world")) world = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--transport")) transport = transport_of(argv[i + 1]);
    }
    if (world < 2) { printf("party-launch: need --world >= 2\n"); return 2; }
    const std::string id = [&] {
        std::vector<uint8_t> b = Net::unique_id(transport);   // RCCL: ncclGetUniqueId here, before any child exists; no GPU work in this process
        return hex(b.data(), b.size());
    }();
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        const pid_t pid = fork();
        if (pid == 0) {
            std::vector<std::string> a = {argv[0], "party", "--rank", std::to_string(r), "--id", id};
            for (int i = 2; i < argc; i++) a.push_back(argv[i]);
            std::vector<char*> av;
            for (auto& x : a) av.push_back(&x[0]);
            av.push_back(nullptr);
            execv(argv[0], av.data());
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    return rc;
}

// `host_demo.bin inputs SEED N` (no GPU needed): the harness's own input generation -- SplitMix64 stream -> canonical values,
// their Montgomery form, the square and the difference of neighbours with the harness's host field code -- for
// tests/test_device_handles.py to compare with tests/util.py and the checker.
static int dump_inputs(int argc, char** argv) {
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    const size_t n = argc > 3 ? (size_t)atoll(argv[3]) : 8;
    std::vector<Fr> c = g16::rand_fr_canonical(seed, n);
    for (size_t i = 0; i < n; i++) {
        const Fr m = g16::hostfr::from_repr(c[i]), m2 = g16::hostfr::from_repr(c[(i + 1) % n]);
        print_fr("canonical", c[i]);
        print_fr("mont", m);
        print_fr("square", g16::hostfr::mont_mul(m, m));
        print_fr("sub", g16::hostfr::sub(m, m2));
        print_fr("add", g16::hostfr::add(m, m2));
    }
    return 0;
}

// `host_demo.bin plonk|marlin [--log-n K | --constraints N] [--parties P] [--steps S] [--warmup W] [--inflight F] [--arena-gb G] [--dump FILE]`:
// BASELINE configs[2] / [3] (mpc-plonk's prover on GSZ lanes; Marlin's AHP rounds, commitments and batched openings on SPDZ lanes) from this
// process alone -- tools/polyvm_host.hpp over include/czk.h, no torch, no Python: the synthetic circuit / index goes up once, S proofs run with F
// in flight (one czk context and host thread each, sharing the registered SRS), commitments / evaluations / opening proofs come down.  Prints one
// JSON line; --dump writes the last proof's outputs in the canonical text form tests/test_pipelines.py compares with the Python host's.
static int polyiop(const char* workload, int argc, char** argv) {
    const bool plonk = !strcmp(workload, "plonk");
    size_t n = (size_t)1 << (plonk ? 18 : 20), parties = plonk ? 3 : 2, steps = 8, warmup = 2, inflight = 4;
    double arena_gb = 0;
    const char* dump = nullptr;
    int rank = -1, world = 0, transport = CZK_NET_SHM, device = 0;
    bool commit_opens = true, breakdown = false;
    double stagger_ms = 0;   // several proofs in flight: prover k starts k * stagger_ms late, so that the provers are not all in the same round at the same time
    std::vector<std::pair<std::string, long>> ctx_options;
    std::vector<uint8_t> id;
    for (int i = 2; i < argc; i++) {
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (!strcmp(argv[i], "--log-n")) n = (size_t)1 << atoi(val());
        else if (!strcmp(argv[i], "--no-commit-opens")) commit_opens = false;
        else if (!strcmp(argv[i], "--breakdown")) breakdown = true;
        else if (!strcmp(argv[i], "--stagger-ms")) stagger_ms = atof(val());
        else if (!strcmp(argv[i], "--ctx-option")) {   // NAME=VALUE: czk_ctx_set_option on every context before its first MSM (repeatable)
            const std::string kv = val();
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) { printf("--ctx-option wants NAME=VALUE\n"); return 2; }
            ctx_options.emplace_back(kv.substr(0, eq), atol(kv.c_str() + eq + 1));
        }
        else if (!strcmp(argv[i], "--rank")) rank = atoi(val());
        else if (!strcmp(argv[i], "--world")) world = atoi(val());
        else if (!strcmp(argv[i], "--id")) id = unhex(val());
        else if (!strcmp(argv[i], "--device")) device = atoi(val());
        else if (!strcmp(argv[i], "--transport")) transport = transport_of(val());
        else if (!strcmp(argv[i], "--constraints")) n = (size_t)atoll(val());
        else if (!strcmp(argv[i], "--parties")) parties = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--steps")) steps = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--warmup")) warmup = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--inflight")) inflight = (size_t)atoi(val());
        else if (!strcmp(argv[i], "--arena-gb")) arena_gb = atof(val());
        else if (!strcmp(argv[i], "--dump")) dump = val();
        else { printf("unknown argument %s\n", argv[i]); return 2; }
    }
    if (steps < 1 || inflight < 1 || parties < 1) { printf("%s: bad arguments\n", workload); return 2; }
    // --world W: the reference's own layout, one process per party (parties = W), evaluations opened through czk::Net.  Without --rank this
    // process only launches the W ranks (fork + exec with a fresh communicator id); rank r writes its dump to FILE.rank<r>.
    const bool party = world > 0;
    if (party && rank < 0) {
        std::vector<uint8_t> b = Net::unique_id(transport);
        const std::string hid = hex(b.data(), b.size());
        std::vector<pid_t> kids;
        for (int r = 0; r < world; r++) {
            const pid_t pid = fork();
            if (pid == 0) {
                std::vector<std::string> a(argv, argv + argc);
                a.insert(a.end(), {"--rank", std::to_string(r), "--id", hid});
                std::vector<char*> av;
                for (auto& x : a) av.push_back(&x[0]);
                av.push_back(nullptr);
                execv(argv[0], av.data());
                _exit(127);
            }
            kids.push_back(pid);
        }
        int rc = 0;
        for (pid_t k : kids) {
            int st = 0;
            waitpid(k, &st, 0);
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
        }
        return rc;
    }
    if (party) {
        if (rank >= world || id.empty() || world < 2) { printf("%s: --world needs >= 2 ranks, --rank < world and --id\n", workload); return 2; }
        parties = (size_t)world;
        inflight = 1;
        if (transport == CZK_NET_RCCL) device = rank;
    }
    inflight = std::min(inflight, steps);
    // before HIP initialises: every context has four streams, so several proofs in flight need more hardware queues than the library's default
    // of 8 (which its loader has already put into the environment: overwrite it)
    if (inflight > 1) setenv("GPU_MAX_HW_QUEUES", "24", 1);
    const size_t per_party = plonk ? 1 : 2;                         // GSZ: one lane per party; SPDZ: sh + mac
    const size_t lanes = party ? per_party : per_party * parties;
    std::vector<int> lift(lanes, plonk ? 1 : 0);                    // public addends: every GSZ lane; the king's sh and mac lanes
    if (!plonk && (!party || rank == 0)) lift[0] = lift[1] = 1;
    const size_t max_deg = plonk ? pvm::plonk_max_degree(n) : pvm::marlin_max_degree(n);
    // arena: the live arrays of one proof peak at ~40 (plonk) / ~60 (marlin) arrays of the longest polynomial on every lane
    const size_t arena_elems = arena_gb > 0 ? (size_t)(arena_gb * 1e9 / 32) : (size_t)(plonk ? 48 : 64) * lanes * (max_deg + 1) + (1 << 20);
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    struct Prover {
        std::unique_ptr<Context> ctx;
        std::unique_ptr<pvm::Machine> B;
        pvm::PlonkInputs pin;
        pvm::MarlinInputs min;
        pvm::Output last;
    };
    std::vector<Prover> provers(inflight);
    auto t_setup = clk::now();
    std::unique_ptr<Net> net;
    for (size_t k = 0; k < inflight; k++) {
        provers[k].ctx.reset(new Context(device));
        for (auto& o : ctx_options) provers[k].ctx->check(czk_ctx_set_option(provers[k].ctx->raw(), o.first.c_str(), o.second));
        provers[k].B.reset(new pvm::Machine(*provers[k].ctx, lanes, max_deg, lift, arena_elems, k ? provers[0].B->srs : nullptr));
        if (party) {
            net.reset(new Net(*provers[0].ctx, transport, rank, world, id));
            pvm::Machine& M = *provers[0].B;
            M.net = net.get();
            M.net_gsz = plonk;
            M.commit_opens = commit_opens;
            M.gsz_degree = (unsigned)((parties - 1) / 2);              // t = (n - 1) / 2 (share/gsz20/mod.rs:94-96)
            M.mac_share = rank == 0 ? pvm::fr_one() : pvm::fr_zero();   // mac_share() = 1 on the king, 0 elsewhere (share/spdz.rs:30-37)
        }
        if (k == 0) provers[0].B->prepare(plonk ? pvm::plonk_commit_sizes(n) : pvm::marlin_commit_sizes(n));
        if (plonk) provers[k].pin = pvm::plonk_inputs(*provers[k].B, n);
        else provers[k].min = pvm::marlin_inputs(*provers[k].B, n);
        provers[k].ctx->sync();
    }
    const double setup_s = secs(t_setup, clk::now());
    auto prove = [&](Prover& p) { p.last = plonk ? pvm::plonk_prove(*p.B, p.pin) : pvm::marlin_prove(*p.B, p.min); };
    auto run = [&](const std::vector<size_t>& share) {
        std::vector<std::thread> th;
        std::vector<std::string> errs(share.size());
        for (size_t k = 0; k < share.size(); k++)
            th.emplace_back([&, k] {
                try {
                    if (stagger_ms > 0 && k) std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(stagger_ms * k));
                    for (size_t i = 0; i < share[k]; i++) prove(provers[k]);
                } catch (const Panic& e) {
                    errs[k] = e.what();
                }
            });
        for (auto& t : th) t.join();
        for (auto& e : errs)
            if (!e.empty()) throw Panic(CZK_ERR_HIP, e);
    };
    auto t0 = clk::now();
    prove(provers[0]);
    const double first_ms = secs(t0, clk::now()) * 1e3;
    run(std::vector<size_t>(inflight, std::max<size_t>(1, warmup ? warmup - 1 : 0)));
    t0 = clk::now();
    prove(provers[0]);                                             // one proof alone
    const double alone_ms = secs(t0, clk::now()) * 1e3;
    // --breakdown: ONE more proof alone, with the accumulate kernels' HIP-event intervals (czk_profile_intervals) laid over the host times at which the
    // transcript points' waits returned: per round (from one challenge to the next) the time an accumulate kernel was running, the head (challenge ->
    // first accumulate kernel: witness polynomials, digit sort), the stalls between accumulate kernels (the next commitment's polynomial was not ready)
    // and the tail (last accumulate kernel -> results on the host: bucket reduction, copy, conversion to affine)
    std::string breakdown_json;
    if (breakdown) {
        Context& c = *provers[0].ctx;
        std::vector<clk::time_point> log;
        provers[0].B->settle_log = &log;
        c.sync();
        c.check(czk_profile_reset(c.raw()));
        c.check(czk_profile_enable(c.raw(), 1));
        const auto base = clk::now();
        prove(provers[0]);
        c.sync();
        const auto fin = clk::now();
        c.check(czk_profile_enable(c.raw(), 0));
        provers[0].B->settle_log = nullptr;
        size_t n_iv = 0;
        std::vector<double> s0(4096), s1(4096);
        c.check(czk_profile_intervals(c.raw(), "msm_accumulate_g1", s0.data(), s1.data(), s0.size(), &n_iv));
        std::vector<std::pair<double, double>> iv;
        for (size_t i = 0; i < n_iv && i < s0.size(); i++) iv.emplace_back(s0[i], s1[i]);
        std::sort(iv.begin(), iv.end());
        static const char* PLONK_TP[] = {"p, pub_q committed", "2 openings at x_pub, gates_q committed", "5 openings at x_gates", "l1, t, q committed (l2_q enqueued)",
                                         "5 openings at r, l2_q committed", "4 openings at x_wiring"};
        static const char* MARLIN_TP[] = {"round 1: w, z_a, z_b, mask_poly", "round 2: t, g_1, h_1", "round 3: g_2, h_2 (+ g_1's degree-bound opening)",
                                          "evaluations at beta, gamma", "batched openings"};
        double prev = 0, tot_busy = 0, tot_head = 0, tot_stall = 0, tot_tail = 0;
        breakdown_json = "[";
        for (size_t r = 0; r < log.size(); r++) {
            const double end = secs(base, log[r]) * 1e3;
            double busy = 0, first = -1, last = prev, cur = prev;
            size_t launches = 0;
            for (auto& v : iv) {
                const double a = std::max(v.first, prev), b = std::min(v.second, end);
                if (b <= a) continue;
                if (first < 0) first = a;
                if (a > cur) cur = a;
                if (b > cur) busy += b - cur, cur = b;
                last = std::max(last, b);
                launches++;
            }
            const double wall = end - prev, head = first < 0 ? wall : first - prev, tail = first < 0 ? 0 : end - last, stall = wall - busy - head - tail;
            const char* label = plonk ? (r < 6 ? PLONK_TP[r] : "?") : (r < 5 ? MARLIN_TP[r] : "?");
            char buf[512];
            snprintf(buf, sizeof buf, "%s{\"round\": \"%s\", \"wall_ms\": %.2f, \"accumulate_busy_ms\": %.2f, \"head_ms\": %.2f, \"stall_ms\": %.2f, \"tail_ms\": %.2f, \"accumulate_launches\": %zu}",
                     r ? ", " : "", label, wall, busy, head, stall, tail, launches);
            breakdown_json += buf;
            tot_busy += busy, tot_head += head, tot_stall += stall, tot_tail += tail;
            prev = end;
        }
        char buf[384];
        snprintf(buf, sizeof buf, "%s{\"round\": \"total\", \"wall_ms\": %.2f, \"accumulate_busy_ms\": %.2f, \"head_ms\": %.2f, \"stall_ms\": %.2f, \"tail_ms\": %.2f, \"proof_ms\": %.2f}]",
                 log.empty() ? "" : ", ", prev, tot_busy, tot_head, tot_stall, tot_tail, secs(base, fin) * 1e3);
        breakdown_json += buf;
    }
    for (auto& p : provers) p.B->msm_count = p.B->ntt_count = p.B->msm_points = 0;
    std::vector<size_t> share(inflight);
    for (size_t k = 0; k < inflight; k++) share[k] = steps / inflight + (k < steps % inflight ? 1 : 0);
    t0 = clk::now();
    run(share);
    const double dt = secs(t0, clk::now());
    const std::string ref = pvm::dump_output(provers[0].last);
    for (size_t k = 1; k < inflight; k++) REQUIRE(pvm::dump_output(provers[k].last) == ref);   // the in-flight provers run the same deterministic inputs
    uint8_t digest[32];
    czk_sha256(ref.data(), ref.size(), digest);
    if (dump) {
        const std::string path = party ? std::string(dump) + ".rank" + std::to_string(rank) : std::string(dump);
        FILE* f = fopen(path.c_str(), "w");
        REQUIRE(f != nullptr);
        fputs(ref.c_str(), f);
        fclose(f);
    }
    if (party) {
        net->barrier();
        if (rank != 0) {
            provers[0].last = pvm::Output();
            provers[0].pin = pvm::PlonkInputs();
            provers[0].min = pvm::MarlinInputs();
            provers[0].B->net = nullptr;
            net.reset();
            return 0;
        }
    }
    size_t msms = 0, ntts = 0;
    for (auto& p : provers) msms += p.B->msm_count, ntts += p.B->ntt_count;
    printf("{\"harness\": \"tools/host_demo.cpp %s (C++ over include/czk.h: tools/polyvm_host.hpp; no torch, no Python)\", \"workload\": \"%s\", \"constraints\": %zu, "
           "\"parties\": %zu, \"layout\": \"%s\", \"transport\": \"%s\", \"share_lanes\": %zu, \"steps\": %zu, \"warmup\": %zu, \"proofs_in_flight\": %zu, \"ms_per_proof\": %.4f, \"proofs_per_s\": %.5f, "
           "\"latency_ms_single_proof\": %.3f, \"first_proof_ms\": %.3f, \"setup_s\": %.3f, \"msms_per_proof\": %.1f, \"ntt_lanes_per_proof\": %.1f, "
           "\"arena_peak_gb\": %.2f, \"commit_opens\": %s, \"in_flight_provers_equal\": true, \"output_sha256\": \"%s\"%s%s}\n",
           workload, workload, n, parties, party ? "party (one process per party, evaluations opened through czk_net)" : "one process",
           party ? transport_name(transport) : "none", lanes, steps, warmup,
           inflight, dt / steps * 1e3, steps / dt, alone_ms, first_ms, setup_s, (double)msms / steps,
           (double)ntts / steps, provers[0].B->arena_peak_bytes() / 1e9, (party && !plonk) ? (commit_opens ? "true" : "false") : "null", hex(digest, 32).c_str(), breakdown ? ", \"single_proof_breakdown\": " : "", breakdown_json.c_str());
    if (net) {
        provers[0].B->net = nullptr;
        net.reset();   // the communicator goes before its context
    }
    // the machines go before their contexts, the shared SRS last
    for (size_t k = provers.size(); k-- > 1;) {
        provers[k].last = pvm::Output();
        provers[k].pin = pvm::PlonkInputs();
        provers[k].min = pvm::MarlinInputs();
        provers[k].B.reset();
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "inputs") == 0) return dump_inputs(argc, argv);
    if (argc > 1 && (strcmp(argv[1], "plonk") == 0 || strcmp(argv[1], "marlin") == 0)) {
        try {
            return polyiop(argv[1], argc, argv);
        } catch (const Panic& p) {
            printf("FAILED: czk::Panic %d: %s\n", p.code, p.what());
            return 1;
        }
    }
    if (argc > 1 && strcmp(argv[1], "party-launch") == 0) return party_launch(argc, argv);
    if (argc > 1 && strcmp(argv[1], "party") == 0) {
        try {
            return party(argc, argv);
        } catch (const Panic& p) {
            printf("FAILED: czk::Panic %d: %s\n", p.code, p.what());
            return 1;
        }
    }
    Context ctx(0);
    if (argc > 1 && strcmp(argv[1], "dump-lift") == 0) return dump_lift(ctx);
    if (argc > 1 && strcmp(argv[1], "bench") == 0) {
        try {
            return bench(ctx, argc, argv);
        } catch (const Panic& p) {
            printf("FAILED: czk::Panic %d: %s\n", p.code, p.what());
            return 1;
        }
    }
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

    // the same transforms on lanes that live on the GPU (DeviceLanes): coset FFT of two lanes equals the host-vector path,
    // divide_by_vanishing_poly_on_coset_in_place likewise; clone() is a device copy
    {
        DeviceLanes dl(ctx, 2, 16);
        dl.upload(0, x);
        dl.upload(1, x);
        dl.len = x.size();
        dom->coset_fft_in_place(dl);
        REQUIRE(dl.len == 16);
        std::vector<Fr> want = x;
        dom->coset_fft_in_place(want);
        DeviceLanes cp = dl.clone();
        std::vector<Fr> g0 = dl.to_host(0), g1 = cp.to_host(1);
        for (size_t i = 0; i < 16; i++) REQUIRE(eq(g0[i], want[i]) && eq(g1[i], want[i]));
        dom->divide_by_vanishing_poly_on_coset_in_place(dl);
        dom->divide_by_vanishing_poly_on_coset_in_place(want);
        g0 = dl.to_host(1);
        for (size_t i = 0; i < 16; i++) REQUIRE(eq(g0[i], want[i]));
        dom->coset_ifft_in_place(cp);
        g1 = cp.to_host(0);
        for (size_t i = 0; i < 13; i++) REQUIRE(eq(g1[i], x[i]));
        // the non-in-place form (domain/mod.rs:130-134): a shorter source with its own lane stride, left untouched
        DeviceLanes src(ctx, 2, x.size());
        src.upload(0, x);
        src.upload(1, x);
        DeviceLanes out = dom->coset_fft(src);
        std::vector<Fr> want2 = x;
        dom->coset_fft_in_place(want2);
        std::vector<Fr> o1 = out.to_host(1), s0 = src.to_host(0);
        REQUIRE(out.capacity() == 16 && src.capacity() == x.size());
        for (size_t i = 0; i < 16; i++) REQUIRE(eq(o1[i], want2[i]));
        for (size_t i = 0; i < x.size(); i++) REQUIRE(eq(s0[i], x[i]));
    }

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
    // the two-lane form with distinct MAC scalars: mac = sum x[i % 13] * [i + 1] G differs from sh, and sh is unchanged
    {
        SpdzGroupShareG1 g2 = SpdzGroupShareG1::multi_scale_pub_group_lanes(bases, sc);
        G1Projective pair[2] = {g2.sh, g2.mac};
        uint64_t a3[24];
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, pair[0].x.l, 2, a3, i2));
        REQUIRE(memcmp(a3, a2, 96) == 0 && memcmp(a3 + 12, a2, 96) != 0);
        std::vector<Fr> macs(n);
        for (size_t i = 0; i < n; i++) macs[i] = sc[i].mac;
        G1Projective want = G1Affine::multi_scalar_mul(bases, macs);
        uint64_t w3[12];
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, want.x.l, 1, w3, i2));
        REQUIRE(memcmp(a3 + 12, w3, 96) == 0);
    }

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

    // KZG10::open against the defining identity, in the exponent: with powers_of_g[i] = [tau^i] G,
    //   commit(p) - [p(z)] G == [tau - z] * open(p, z).w          (e(C - vG, H) = e(w, (tau - z) H) without the pairing)
    {
        const uint64_t tau = 5, zpt = 2;
        const size_t deg1 = 24;                                    // 5^23 < 2^64
        std::vector<uint64_t> pw(4 * (deg1 + 1), 0);
        uint64_t t = 1;
        for (size_t i = 0; i < deg1; i++, t *= tau) pw[4 * i] = t;
        pw[4 * deg1] = 1;                                          // G itself
        std::vector<uint64_t> ppts(12 * (deg1 + 1));
        ctx.check(czk_fixed_base_points(ctx.raw(), CZK_G1, pw.data(), deg1 + 1, ppts.data(), CZK_MEM_HOST));
        G1Bases powers(ctx, ppts.data(), nullptr, deg1);
        std::vector<uint64_t> craw;
        for (size_t i = 0; i < deg1; i++) craw.push_back(1000003 * i + 17);
        std::vector<Fr> poly = fr_from_u64(ctx, craw);
        Fr zf = fr_from_u64(ctx, {zpt})[0], value;
        std::vector<Fr> wit = KZG10::compute_witness_polynomial(ctx, poly, zf, &value);
        REQUIRE(wit.size() == deg1 - 1);
        KZG10::Proof proof = KZG10::open(powers, poly, zf);
        REQUIRE(!proof.random_v.has_value());
        G1Projective cm = KZG10::commit(powers, poly);
        // lhs = cm + [-value] G : one-point MSM with the negated Montgomery scalar
        Fr zero{{0, 0, 0, 0}}, negv;
        ctx.check(czk_fr_vec_op(ctx.raw(), CZK_OP_SUB, zero.l, value.l, negv.l, 1, CZK_MEM_HOST));
        uint8_t no_inf = 0;
        G1Projective vg;
        ctx.check(czk_msm_g1(ctx.raw(), &ppts[12 * deg1], &no_inf, negv.l, 1, 1, CZK_SCALAR_MONTGOMERY, vg.x.l));
        G1Projective lhs;
        ctx.check(czk_jac_add(ctx.raw(), CZK_G1, cm.x.l, vg.x.l, lhs.x.l));
        // rhs = [tau - z] w
        uint64_t waff[12], laff[12], raff[12];
        uint8_t winf = 0, linf = 0, rinf = 0;
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, proof.w.x.l, 1, waff, &winf));
        REQUIRE(!winf);
        uint64_t kk[4] = {tau - zpt, 0, 0, 0};
        G1Projective rhs;
        ctx.check(czk_msm_g1(ctx.raw(), waff, &no_inf, kk, 1, 1, CZK_SCALAR_CANONICAL, rhs.x.l));
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, lhs.x.l, 1, laff, &linf));
        ctx.check(czk_jac_to_affine(ctx.raw(), CZK_G1, rhs.x.l, 1, raff, &rinf));
        REQUIRE(!linf && !rinf && memcmp(laff, raff, 96) == 0);
    }

    // ConstraintMatrix::evaluate: rows {z0 + 3 z2, (empty), z1} on z = (2, 5, 7)
    {
        std::vector<Fr> c = fr_from_u64(ctx, {1, 3, 2, 5, 7, 23});
        std::vector<std::vector<std::pair<Fr, size_t>>> rows = {{{c[0], 0}, {c[1], 2}}, {}, {{c[0], 1}}};
        ConstraintMatrix mat(ctx, rows, 3);
        std::vector<Fr> ev = mat.evaluate({c[2], c[3], c[4]}, 4);
        REQUIRE(ev.size() == 4 && eq(ev[0], c[5]) && eq(ev[1], Fr{{0, 0, 0, 0}}) && eq(ev[2], c[3]) && eq(ev[3], Fr{{0, 0, 0, 0}}));
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

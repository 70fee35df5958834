// net.hip -- czk_net_*: the transport of the MPC opens behind the C ABI, and the reference's batch opens
// as single calls on device lanes.  Host code only (the kernels it launches live in share.hip / ntt.hip).
//
//   reference                                                            here
//   MpcMultiNet::broadcast          mpc-net/src/multi.rs:145-173          czk_net_broadcast       (RCCL: ncclAllGather | grouped ncclSend / ncclRecv)
//   MpcMultiNet::send_to_king       mpc-net/src/multi.rs:175-210          czk_net_send_to_king    (RCCL: grouped ncclSend -> king's ncclRecv)
//   MpcMultiNet::recv_from_king     mpc-net/src/multi.rs:211-242          czk_net_recv_from_king
//   MpcSerNet::atomic_broadcast     mpc-algebra/src/channel.rs:50-75      czk_net_atomic_broadcast
//   SpdzFieldShare::batch_open      mpc-algebra/src/share/spdz.rs:166-185 czk_spdz_batch_open
//   AdditiveFieldShare::batch_open  mpc-algebra/src/share/add.rs:256-259  czk_add_batch_open
//   GszFieldShare::batch_open       share/gsz20/mod.rs:286-300, 440-466   czk_gsz_batch_open
//   gsz20::batch_king_compute       share/gsz20/mod.rs:494-527            czk_gsz_batch_king_compute
//   Vec<Fr> wire format             algebra/serialize/src/lib.rs:220-229  czk_fr_vec_serialize / _deserialize
//
// Two transports (include/czk.h): RCCL for one party per GPU -- every exchange is enqueued on the context's stream, so the kernels that
// produce a share lane, the exchange and the kernels that consume the gathered lanes are ordered by the stream itself, no host
// synchronisation and no copy out of HBM; and SHM for parties that are processes of one node in any assignment to GPUs (several on
// one GPU), staged through a pinned POSIX shared-memory segment with a generation barrier -- what the party layout is tested with
// on a one-GPU box.  RCCL is resolved with dlopen at the first RCCL communicator: libczk_hip.so does not link against it.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>   // types and prototypes only
#include <sched.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cstring>

#include "czk_internal.h"

using namespace czk;

// ---- SHA-256 (FIPS 180-4): the reference's CommitHash (mpc-algebra/src/channel.rs:92) ---------------------------------------------
namespace {
struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    Sha256() {
        static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(h, iv, sizeof h);
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g, g = f, f = e, e = d + t1, d = c, c = b, b = a, a = t1 + t2;
        }
        h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e, h[5] += f, h[6] += g, h[7] += hh;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        len += n;
        if (fill) {
            const size_t take = n < 64 - fill ? n : 64 - fill;
            memcpy(buf + fill, p, take);
            fill += take, p += take, n -= take;
            if (fill < 64) return;
            block(buf);
            fill = 0;
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) memcpy(buf, p, n), fill = n;
    }
    void finish(uint8_t* out32) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t be[8];
        for (int i = 0; i < 8; i++) be[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(be, 8);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 4; j++) out32[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
    }
};

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds an RCCL (PyTorch-ROCm bundles one under the same SONAME) gets that copy back: one RCCL per process
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.err = std::string("dlopen librccl.so.1: ") + (e ? e : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p) ok = false, r.err = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) r.handle = nullptr;
    });
    return &r;
}

// ---- SHM control block (lives in the shared segment; every field is an address-free lock-free atomic on x86-64) ------------------
struct ShmHeader {
    std::atomic<uint64_t> magic;
    std::atomic<uint32_t> world;
    std::atomic<uint32_t> reserved;  // (keeps the layout of the control block)
    std::atomic<uint32_t> count;     // arrivals at the current barrier
    std::atomic<uint32_t> abort;     // a rank timed out or failed: everybody leaves with CZK_ERR_NET
    std::atomic<uint64_t> gen;       // barrier generation
    std::atomic<uint64_t> slot_bytes;   // published by rank 0 when it creates the data segment
};
constexpr uint64_t SHM_MAGIC = 0x314e4b5a43ull;   // "CZKN1"
}  // namespace

struct czk_net {
    czk_ctx* ctx = nullptr;
    int transport = 0, rank = 0, world = 1;
    std::string err;
    long exchange = 0;            // 0 ring, 1 p2p
    long timeout_ms = 120000;
    uint64_t stats[5] = {0, 0, 0, 0, 0};
    // RCCL
    ncclComm_t comm = nullptr;
    // SHM
    std::string shm_name;
    ShmHeader* hdr = nullptr;
    char* slots = nullptr;        // world x 2 x slot_bytes
    size_t slot_bytes = (size_t)16 << 20, data_bytes = 0;
    bool data_pinned = false;
    // IPC (the SHM control block + one device mailbox per rank, mapped into every peer with hipIpc: staging never leaves device memory)
    char* mailbox = nullptr;                 // this rank's 2 x slot_bytes, device memory
    std::vector<char*> peer_mail;            // every rank's mailbox as this process sees it (own entry = mailbox)
    // lab build, IPC transport created WITHOUT a context: a dry run of the mailbox hand-over on a box without GPUs (tests/test_net.py) -- every rank's "device
    // mailbox" is a POSIX segment of its own, its 64-byte handle names the owner's rank and device (= rank: every peer is on another device), and the order
    // allocate -> publish handle -> barrier -> open every peer's handle -> barrier -> exchanges by slot parity is the code the hipIpc path runs
    bool ipc_dry = false;
    size_t dry_mail_bytes = 0;
    bool shm_like() const { return transport == CZK_NET_SHM || transport == CZK_NET_IPC; }
    uint64_t seq = 0;             // chunk steps so far: parity of the slot in use
    bool reads_in_flight = false;
    // scratch on the context's GPU
    DeviceBuf gather, dx, small;
};

namespace {
int net_err(czk_net* n, int code, const std::string& msg) {
    if (n) n->err = msg;
    return code;
}
#define NET_HIP(n, call)                                                                                  \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess) return net_err((n), CZK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define NET_NCCL(n, call)                                                                                 \
    do {                                                                                                  \
        ncclResult_t e__ = (call);                                                                        \
        if (e__ != ncclSuccess) return net_err((n), CZK_ERR_NET, std::string(#call) + ": " + rccl()->GetErrorString(e__)); \
    } while (0)
// an error of a czk_* call on the communicator's context becomes the communicator's error too
#define NET_CTX(n, expr)                                                      \
    do {                                                                      \
        int rc__ = (expr);                                                    \
        if (rc__ != CZK_OK) return net_err((n), rc__, czk_last_error((n)->ctx)); \
    } while (0)

int net_buf(czk_net* n, DeviceBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return CZK_OK;
    if (b.p) {
        NET_HIP(n, hipStreamSynchronize(n->ctx->stream));   // earlier exchanges may still read the old allocation
        (void)hipFree(b.p);
        b.p = nullptr, b.bytes = 0;
    }
    if (hipMalloc(&b.p, bytes) != hipSuccess) return net_err(n, CZK_ERR_NOMEM, "hipMalloc communicator scratch");
    b.bytes = bytes;
    return CZK_OK;
}

// ---- SHM transport -------------------------------------------------------------------------------------------------------------------
int shm_barrier(czk_net* n) {
    ShmHeader* h = n->hdr;
    if (n->ctx) chaos_point(n->ctx, n->ctx->stream);   // lab option "chaos": the parties reach every generation flip at perturbed times (no-op otherwise)
    if (h->abort.load(std::memory_order_acquire)) return net_err(n, CZK_ERR_NET, "shm: a peer aborted the communicator");
    const uint64_t gen = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)n->world) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.store(gen + 1, std::memory_order_release);
        return CZK_OK;
    }
    const double t0 = now_ms();
    for (unsigned spin = 0;; spin++) {
        if (h->gen.load(std::memory_order_acquire) != gen) return CZK_OK;
        if (h->abort.load(std::memory_order_acquire)) return net_err(n, CZK_ERR_NET, "shm: a peer aborted the communicator");
        if (spin < 2000) continue;
        if ((spin & 63) == 0 && now_ms() - t0 > (double)n->timeout_ms) {
            h->abort.store(1, std::memory_order_release);
            return net_err(n, CZK_ERR_NET, "shm: a peer did not arrive within timeout_ms (ranks must issue the same sequence of exchanges)");
        }
        if (spin < 20000) sched_yield();
        else usleep(50);
    }
}

std::string shm_name_of(const uint8_t* id, size_t len, const char* suffix) {
    static const char* hex = "0123456789abcdef";
    std::string s = "/czk_net_";
    for (size_t i = 0; i < len; i++) s += hex[id[i] >> 4], s += hex[id[i] & 15];
    return s + suffix;
}

// maps `bytes` of the named segment; rank 0 creates it (replacing a stale one), the others wait for it to appear at full size
int shm_map(czk_net* n, const std::string& name, size_t bytes, bool create, void** out) {
    int fd = -1;
    if (create) {
        shm_unlink(name.c_str());
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return net_err(n, CZK_ERR_NET, "shm_open(create) " + name + ": " + strerror(errno));
        if (ftruncate(fd, (off_t)bytes) != 0) {
            const std::string e = strerror(errno);
            close(fd);
            shm_unlink(name.c_str());
            return net_err(n, CZK_ERR_NET, "ftruncate " + name + ": " + e);
        }
    } else {
        const double t0 = now_ms();
        for (;;) {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            if (fd >= 0) close(fd), fd = -1;
            if (now_ms() - t0 > (double)n->timeout_ms) return net_err(n, CZK_ERR_NET, "shm: rank 0 never created " + name);
            usleep(200);
        }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return net_err(n, CZK_ERR_NET, "mmap " + name + ": " + strerror(errno));
    *out = p;
    return CZK_OK;
}

// IPC transport: every rank allocates its mailbox on its GPU, publishes the hipIpc handle through a small shared segment and maps the others'
int ipc_data(czk_net* n) {
    if (!n->peer_mail.empty()) return CZK_OK;
    const std::string name = n->shm_name + ".h";
    const size_t hb = (size_t)n->world * sizeof(hipIpcMemHandle_t);
    if (n->rank == 0) n->hdr->slot_bytes.store(n->slot_bytes, std::memory_order_release);
    void* p = nullptr;
    int rc = CZK_OK;
    if (n->rank == 0) rc = shm_map(n, name, hb, true, &p);
    if (rc != CZK_OK) n->hdr->abort.store(1, std::memory_order_release);
    CZK_TRY(rc);
    CZK_TRY(shm_barrier(n));
    if (n->rank != 0) {
        n->slot_bytes = (size_t)n->hdr->slot_bytes.load(std::memory_order_acquire);
        rc = shm_map(n, name, hb, false, &p);
        if (rc != CZK_OK) n->hdr->abort.store(1, std::memory_order_release);
        CZK_TRY(rc);
    }
    hipIpcMemHandle_t* handles = (hipIpcMemHandle_t*)p;
    auto fail = [&](const std::string& msg) {
        n->hdr->abort.store(1, std::memory_order_release);
        munmap(p, hb);
        return net_err(n, CZK_ERR_HIP, msg);
    };
    // the three steps that touch HIP, with their dry-run stand-ins (lab build, no context): allocate this rank's mailbox, make its handle, open a peer's handle
    struct DryHandle {
        uint64_t magic;
        int32_t rank, device;
    };
    constexpr uint64_t DRY_MAGIC = 0x59524449414d5a43ull;   // "CZMAIDRY"
    static_assert(sizeof(DryHandle) <= sizeof(hipIpcMemHandle_t), "a dry-run handle fits the slot of a real one");
    auto mail_name = [&](int r) { return n->shm_name + ".m" + std::to_string(r); };
    void* mb = nullptr;
    if (n->ipc_dry) {
        n->dry_mail_bytes = 2 * n->slot_bytes;
        if (shm_map(n, mail_name(n->rank), n->dry_mail_bytes, true, &mb) != CZK_OK) return fail("dry run: mailbox segment: " + n->err);
        memset(&handles[n->rank], 0, sizeof(hipIpcMemHandle_t));
        DryHandle dh{DRY_MAGIC, n->rank, n->rank};
        memcpy(&handles[n->rank], &dh, sizeof dh);
    } else {
        if (hipMalloc(&mb, 2 * n->slot_bytes) != hipSuccess) return fail("hipMalloc mailbox");
        hipError_t e = hipIpcGetMemHandle(&handles[n->rank], mb);
        if (e != hipSuccess) {
            (void)hipFree(mb);
            return fail(std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment?)");
        }
    }
    n->mailbox = (char*)mb;
    CZK_TRY(shm_barrier(n));   // every handle is published
    std::vector<char*> peers((size_t)n->world, nullptr);
    for (int r = 0; r < n->world; r++) {
        if (r == n->rank) {
            peers[r] = n->mailbox;
            continue;
        }
        void* q = nullptr;
        if (n->ipc_dry) {
            DryHandle dh;
            memcpy(&dh, &handles[r], sizeof dh);
            if (dh.magic != DRY_MAGIC || dh.rank != r || dh.device == n->rank) return fail("dry run: slot " + std::to_string(r) + " of the handle table does not hold rank " + std::to_string(r) + "'s handle");
            if (shm_map(n, mail_name(r), n->dry_mail_bytes, false, &q) != CZK_OK) return fail("dry run: open mailbox of rank " + std::to_string(r) + ": " + n->err);
        } else {
            hipError_t e = hipIpcOpenMemHandle(&q, handles[r], hipIpcMemLazyEnablePeerAccess);   // (a peer on another GPU: peer access over xGMI, enabled on first use)
            if (e != hipSuccess) return fail(std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
        }
        peers[r] = (char*)q;
    }
    CZK_TRY(shm_barrier(n));   // everybody has every mailbox mapped
    if (n->ipc_dry) shm_unlink(mail_name(n->rank).c_str());   // (mapped by every peer: the name can go)
    if (n->rank == 0) shm_unlink(name.c_str());
    munmap(p, hb);
    n->peer_mail = peers;
    return CZK_OK;
}

int shm_data(czk_net* n) {   // the staging slots, created at the first exchange (so "slot_bytes" can be set after czk_net_create)
    if (n->transport == CZK_NET_IPC) return ipc_data(n);
    if (n->slots) return CZK_OK;
    const std::string name = n->shm_name + ".d";
    if (n->rank == 0) n->hdr->slot_bytes.store(n->slot_bytes, std::memory_order_release);
    void* p = nullptr;
    int rc = CZK_OK;
    if (n->rank == 0) rc = shm_map(n, name, (size_t)n->world * 2 * n->slot_bytes, true, &p);
    if (rc != CZK_OK) n->hdr->abort.store(1, std::memory_order_release);
    CZK_TRY(rc);
    CZK_TRY(shm_barrier(n));   // rank 0 has created the segment and published its slot size
    if (n->rank != 0) {
        n->slot_bytes = (size_t)n->hdr->slot_bytes.load(std::memory_order_acquire);
        rc = shm_map(n, name, (size_t)n->world * 2 * n->slot_bytes, false, &p);
        if (rc != CZK_OK) n->hdr->abort.store(1, std::memory_order_release);
        CZK_TRY(rc);
    }
    n->data_bytes = (size_t)n->world * 2 * n->slot_bytes;
    CZK_TRY(shm_barrier(n));   // everybody has it mapped
    if (n->rank == 0) shm_unlink(name.c_str());
    if (n->ctx) {              // pinned: the DMA engines read / write the segment directly
        (void)hipSetDevice(n->ctx->device);
        n->data_pinned = hipHostRegister(p, n->data_bytes, hipHostRegisterPortable) == hipSuccess;
        if (!n->data_pinned) (void)hipGetLastError();   // pageable staging still works, only slower
    }
    n->slots = (char*)p;
    return CZK_OK;
}

inline char* shm_slot(czk_net* n, int owner, uint64_t parity) {
    if (n->transport == CZK_NET_IPC) return n->peer_mail[(size_t)owner] + (parity & 1) * n->slot_bytes;
    return n->slots + ((size_t)owner * 2 + (parity & 1)) * n->slot_bytes;
}

int shm_put(czk_net* n, char* slot, const void* src, size_t len, int mem, bool* wrote) {
    if (n->ctx) chaos_point(n->ctx, n->ctx->stream);
    if (n->transport == CZK_NET_IPC && !n->ipc_dry) {   // the slot is device memory (this rank's mailbox, or a peer's through its mapping)
        NET_HIP(n, hipMemcpyAsync(slot, src, len, mem == CZK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, n->ctx->stream));
        *wrote = true;
        return CZK_OK;
    }
    if (mem == CZK_MEM_DEVICE) {
        NET_HIP(n, hipMemcpyAsync(slot, src, len, hipMemcpyDeviceToHost, n->ctx->stream));
        *wrote = true;
    } else {
        memcpy(slot, src, len);
    }
    return CZK_OK;
}
int shm_get(czk_net* n, void* dst, const char* slot, size_t len, int mem) {
    if (n->ctx) chaos_point(n->ctx, n->ctx->stream);
    if (n->transport == CZK_NET_IPC && !n->ipc_dry) {
        NET_HIP(n, hipMemcpyAsync(dst, slot, len, mem == CZK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, n->ctx->stream));
        if (mem == CZK_MEM_DEVICE) n->reads_in_flight = true;
        else NET_HIP(n, hipStreamSynchronize(n->ctx->stream));   // host destination: complete before the call returns
        return CZK_OK;
    }
    if (mem == CZK_MEM_DEVICE) {
        NET_HIP(n, hipMemcpyAsync(dst, slot, len, hipMemcpyHostToDevice, n->ctx->stream));
        n->reads_in_flight = true;
    } else {
        memcpy(dst, slot, len);
    }
    return CZK_OK;
}
int local_copy(czk_net* n, void* dst, const void* src, size_t len, int mem) {
    if (dst == src || !len) return CZK_OK;
    if (mem == CZK_MEM_DEVICE) NET_HIP(n, hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToDevice, n->ctx->stream));
    else memmove(dst, src, len);
    return CZK_OK;
}
// every rank arrives here once per chunk step: its slot writes (and the reads of the step before) have completed
int shm_step_barrier(czk_net* n, bool wrote) {
    if (n->ctx && (wrote || n->reads_in_flight)) {
        NET_HIP(n, hipStreamSynchronize(n->ctx->stream));
        n->reads_in_flight = false;
    }
    return shm_barrier(n);
}

enum class Op { Broadcast, ToKing, FromKing };

// One exchange in chunk steps of slot_bytes.  Step k uses the slots of parity k: a slot written in step k is read in step k and
// next written in step k + 2, which lies behind barrier k + 1 -- and a rank arrives there only after its reads of step k completed.
int shm_exchange(czk_net* n, Op op, const char* send, size_t bytes, char* recv, int mem) {
    CZK_TRY(shm_data(n));
    const int W = n->world, me = n->rank;
    if (op == Op::Broadcast) CZK_TRY(local_copy(n, recv + (size_t)me * bytes, send, bytes, mem));
    if (op == Op::ToKing && me == 0) CZK_TRY(local_copy(n, recv, send, bytes, mem));
    if (op == Op::FromKing && me == 0) CZK_TRY(local_copy(n, recv, send, bytes, mem));
    for (size_t off = 0; off < bytes || off == 0; off += n->slot_bytes) {
        const size_t len = bytes - off < n->slot_bytes ? bytes - off : n->slot_bytes;
        const uint64_t par = n->seq++;
        bool wrote = false;
        if (len) {
            if (op == Op::Broadcast || (op == Op::ToKing && me != 0)) CZK_TRY(shm_put(n, shm_slot(n, me, par), send + off, len, mem, &wrote));
            if (op == Op::FromKing && me == 0)
                for (int p = 1; p < W; p++) CZK_TRY(shm_put(n, shm_slot(n, p, par), send + (size_t)p * bytes + off, len, mem, &wrote));
        }
        CZK_TRY(shm_step_barrier(n, wrote));
        if (len) {
            if (op == Op::Broadcast || (op == Op::ToKing && me == 0)) {
                for (int p = 0; p < W; p++)
                    if (p != me) CZK_TRY(shm_get(n, recv + (size_t)p * bytes + off, shm_slot(n, p, par), len, mem));
            }
            if (op == Op::FromKing && me != 0) CZK_TRY(shm_get(n, recv + off, shm_slot(n, me, par), len, mem));
        }
        if (bytes == 0) break;
    }
    return CZK_OK;
}

// ---- RCCL transport --------------------------------------------------------------------------------------------------------------------
int rccl_exchange_dev(czk_net* n, Op op, const char* send, size_t bytes, char* recv) {
    Rccl* R = rccl();
    hipStream_t st = n->ctx->stream;
    const int W = n->world, me = n->rank;
    if (bytes == 0) return CZK_OK;
    if (op == Op::Broadcast && n->exchange == 0) {
        NET_NCCL(n, R->AllGather(send, recv, bytes, ncclChar, n->comm, st));
        return CZK_OK;
    }
    if (op == Op::Broadcast) {   // p2p: world - 1 concurrent copies of this party's buffer, one per xGMI link
        CZK_TRY(local_copy(n, recv + (size_t)me * bytes, send, bytes, CZK_MEM_DEVICE));
        NET_NCCL(n, R->GroupStart());
        for (int d = 1; d < W; d++) {
            const int to = (me + d) % W, from = (me - d + W) % W;
            NET_NCCL(n, R->Send(send, bytes, ncclChar, to, n->comm, st));
            NET_NCCL(n, R->Recv(recv + (size_t)from * bytes, bytes, ncclChar, from, n->comm, st));
        }
        NET_NCCL(n, R->GroupEnd());
        return CZK_OK;
    }
    if (me == 0) CZK_TRY(local_copy(n, recv, send, bytes, CZK_MEM_DEVICE));
    if (W == 1) return CZK_OK;
    NET_NCCL(n, R->GroupStart());
    if (op == Op::ToKing) {
        if (me == 0)
            for (int p = 1; p < W; p++) NET_NCCL(n, R->Recv(recv + (size_t)p * bytes, bytes, ncclChar, p, n->comm, st));
        else
            NET_NCCL(n, R->Send(send, bytes, ncclChar, 0, n->comm, st));
    } else {
        if (me == 0)
            for (int p = 1; p < W; p++) NET_NCCL(n, R->Send(send + (size_t)p * bytes, bytes, ncclChar, p, n->comm, st));
        else
            NET_NCCL(n, R->Recv(recv, bytes, ncclChar, 0, n->comm, st));
    }
    NET_NCCL(n, R->GroupEnd());
    return CZK_OK;
}

// sizes of the send / recv side of an exchange on this rank (0 = that side is not touched here)
void op_sizes(const czk_net* n, Op op, size_t bytes, size_t* send_b, size_t* recv_b) {
    const bool king = n->rank == 0;
    switch (op) {
        case Op::Broadcast: *send_b = bytes, *recv_b = bytes * n->world; break;
        case Op::ToKing: *send_b = bytes, *recv_b = king ? bytes * n->world : 0; break;
        case Op::FromKing: *send_b = king ? bytes * n->world : 0, *recv_b = bytes; break;
    }
}

int exchange(czk_net* n, Op op, const void* send, size_t bytes, void* recv, int mem) {
    if (!n) return CZK_ERR_ARG;
    if (!valid_mem(mem)) return net_err(n, CZK_ERR_ARG, "czk_net: mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    size_t sb, rb;
    op_sizes(n, op, bytes, &sb, &rb);
    if (bytes && ((sb && !send) || (rb && !recv))) return net_err(n, CZK_ERR_ARG, "czk_net: null buffer");
    if (mem == CZK_MEM_DEVICE && !n->ctx) return net_err(n, CZK_ERR_ARG, "czk_net: device buffers need a communicator created with a context");
    if (n->ctx) NET_HIP(n, hipSetDevice(n->ctx->device));
    if (n->shm_like()) {
        const int rc = shm_exchange(n, op, (const char*)send, bytes, (char*)recv, mem);
        if (rc != CZK_OK) n->hdr->abort.store(1, std::memory_order_release);   // the peers leave their barrier at once instead of after timeout_ms
        return rc;
    }
    if (mem == CZK_MEM_DEVICE) return rccl_exchange_dev(n, op, (const char*)send, bytes, (char*)recv);
    // host buffers over RCCL (commitments, digests): staged through the communicator's device scratch, blocking
    const size_t ro = (sb + 15) & ~(size_t)15;
    CZK_TRY(net_buf(n, n->small, ro + rb + 16));
    char* ds = (char*)n->small.p;
    char* dr = ds + ro;
    if (sb) NET_HIP(n, hipMemcpyAsync(ds, send, sb, hipMemcpyHostToDevice, n->ctx->stream));
    CZK_TRY(rccl_exchange_dev(n, op, ds, bytes, dr));
    if (rb) NET_HIP(n, hipMemcpyAsync(recv, dr, rb, hipMemcpyDeviceToHost, n->ctx->stream));
    NET_HIP(n, hipStreamSynchronize(n->ctx->stream));
    return CZK_OK;
}

void count_stats(czk_net* n, Op op, size_t m) {   // mpc-net/src/multi.rs:148-150, :179-193, :214-220
    const uint64_t others = (uint64_t)(n->world - 1);
    switch (op) {
        case Op::Broadcast:
            n->stats[0] += others * m, n->stats[1] += others * m, n->stats[2]++;
            break;
        case Op::ToKing:
            n->stats[3]++;
            if (n->rank == 0) n->stats[1] += others * m;
            else n->stats[0] += m;
            break;
        case Op::FromKing:
            n->stats[4]++;
            if (n->rank == 0) n->stats[0] += others * (m + 8);
            else n->stats[1] += m;
            break;
    }
}
}  // namespace

// ---- C ABI: communicator ---------------------------------------------------------------------------------------------------------------
extern "C" void czk_sha256(const void* data, size_t len, uint8_t* out32) {
    Sha256 s;
    s.update(data, len);
    s.finish(out32);
}

extern "C" int czk_net_unique_id(int transport, uint8_t* out, size_t cap, size_t* len) {
    if (!out || !len) return CZK_ERR_ARG;
    if (transport == CZK_NET_RCCL) {
        if (cap < NCCL_UNIQUE_ID_BYTES) return CZK_ERR_ARG;
        Rccl* R = rccl();
        if (!R->handle) return CZK_ERR_NET;
        ncclUniqueId id;
        if (R->GetUniqueId(&id) != ncclSuccess) return CZK_ERR_NET;
        memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
        *len = NCCL_UNIQUE_ID_BYTES;
        return CZK_OK;
    }
    if (transport == CZK_NET_SHM || transport == CZK_NET_IPC) {
        if (cap < 16) return CZK_ERR_ARG;
        if (getrandom(out, 16, 0) != 16) return CZK_ERR_NET;
        *len = 16;
        return CZK_OK;
    }
    return CZK_ERR_ARG;
}

extern "C" void czk_net_destroy(czk_net* n) {
    if (!n) return;
    if (n->ctx) {
        (void)hipSetDevice(n->ctx->device);
        (void)hipStreamSynchronize(n->ctx->stream);
    }
    if (n->comm) (void)rccl()->CommDestroy(n->comm);
    if (n->slots) {
        if (n->data_pinned) (void)hipHostUnregister(n->slots);
        munmap(n->slots, n->data_bytes);
    }
    for (size_t r = 0; r < n->peer_mail.size(); r++)
        if ((int)r != n->rank && n->peer_mail[r]) {
            if (n->ipc_dry) munmap(n->peer_mail[r], n->dry_mail_bytes);
            else (void)hipIpcCloseMemHandle(n->peer_mail[r]);
        }
    if (n->mailbox) {   // (a peer that still has it mapped keeps the memory alive until it closes its handle)
        if (n->ipc_dry) munmap(n->mailbox, n->dry_mail_bytes);
        else (void)hipFree(n->mailbox);
    }
    if (n->hdr) munmap(n->hdr, sizeof(ShmHeader));
    for (DeviceBuf* b : {&n->gather, &n->dx, &n->small})
        if (b->p) (void)hipFree(b->p);
    delete n;
}

extern "C" int czk_net_create(czk_ctx* ctx, int transport, int rank, int world, const uint8_t* id, size_t id_len, czk_net** out) {
    if (!out) return CZK_ERR_ARG;
    *out = nullptr;
    auto fail = [&](int code, const std::string& msg) { return ctx ? set_err(ctx, code, msg) : code; };
    if (world < 1 || rank < 0 || rank >= world || !id || !id_len) return fail(CZK_ERR_ARG, "czk_net_create: bad rank / world / id");
    czk_net* n = new czk_net;
    n->ctx = ctx, n->transport = transport, n->rank = rank, n->world = world;
    if (ctx && ctx->net_create_timeout_ms > 0) n->timeout_ms = ctx->net_create_timeout_ms;   // czk_ctx_set_option "net_create_timeout_ms": the rendezvous itself
    int rc = CZK_ERR_ARG;
    if (transport == CZK_NET_RCCL) {
        rc = [&]() -> int {
            if (!ctx) return net_err(n, CZK_ERR_ARG, "czk_net_create: the RCCL transport needs a context");
            if (id_len != NCCL_UNIQUE_ID_BYTES) return net_err(n, CZK_ERR_ARG, "czk_net_create: an RCCL id is the 128 bytes of czk_net_unique_id");
            Rccl* R = rccl();
            if (!R->handle) return net_err(n, CZK_ERR_NET, R->err);
            NET_HIP(n, hipSetDevice(ctx->device));
            ncclUniqueId uid;
            memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
            NET_NCCL(n, R->CommInitRank(&n->comm, world, uid, rank));
            return CZK_OK;
        }();
    } else if (transport == CZK_NET_SHM || transport == CZK_NET_IPC) {
        rc = [&]() -> int {
#ifdef CZK_LAB
            if (transport == CZK_NET_IPC && !ctx) n->ipc_dry = true;   // lab build: the dry run of the mailbox hand-over (see czk_net::ipc_dry); host buffers only
#else
            if (transport == CZK_NET_IPC && !ctx) return net_err(n, CZK_ERR_ARG, "czk_net_create: the IPC transport needs a context");
#endif
            if (id_len > 32) return net_err(n, CZK_ERR_ARG, "czk_net_create: an SHM id is 1..32 bytes");
            n->shm_name = shm_name_of(id, id_len, "");
            void* p = nullptr;
            CZK_TRY(shm_map(n, n->shm_name, sizeof(ShmHeader), rank == 0, &p));
            n->hdr = (ShmHeader*)p;
            if (rank == 0) {   // ftruncate zero-filled the segment: every counter starts at 0; magic last
                n->hdr->world.store((uint32_t)world, std::memory_order_relaxed);
                n->hdr->magic.store(SHM_MAGIC, std::memory_order_release);
            } else {
                const double t0 = now_ms();
                while (n->hdr->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
                    if (now_ms() - t0 > (double)n->timeout_ms) {
                        n->hdr->abort.store(1, std::memory_order_release);
                        return net_err(n, CZK_ERR_NET, "shm: rank 0 never initialised the communicator");
                    }
                    usleep(100);
                }
                if (n->hdr->world.load(std::memory_order_relaxed) != (uint32_t)world) {
                    n->hdr->abort.store(1, std::memory_order_release);   // the ranks already in the barrier leave at once instead of after timeout_ms
                    return net_err(n, CZK_ERR_NET, "shm: ranks disagree on the world size");
                }
            }
            CZK_TRY(shm_barrier(n));   // collective: everybody has the control block mapped
            if (rank == 0) shm_unlink(n->shm_name.c_str());
            return CZK_OK;
        }();
    } else {
        n->err = "czk_net_create: unknown transport";
    }
    if (rc != CZK_OK) {
        const std::string msg = n->err;
        if (n->hdr && rank == 0) shm_unlink(n->shm_name.c_str());
        czk_net_destroy(n);
        return fail(rc, msg);
    }
    *out = n;
    return CZK_OK;
}

extern "C" int czk_net_rank(const czk_net* n) { return n ? n->rank : -1; }
extern "C" int czk_net_world(const czk_net* n) { return n ? n->world : 0; }
extern "C" const char* czk_net_last_error(const czk_net* n) { return n ? n->err.c_str() : "null communicator"; }

extern "C" int czk_net_set_option(czk_net* n, const char* name, long value) {
    if (!n || !name) return CZK_ERR_ARG;
    const std::string k = name;
    if (k == "exchange" && (value == 0 || value == 1)) n->exchange = value;
    else if (k == "timeout_ms" && value > 0) n->timeout_ms = value;
    else if (k == "slot_bytes" && value >= 64 && !n->slots && n->peer_mail.empty()) n->slot_bytes = ((size_t)value + 63) & ~(size_t)63;
    else return net_err(n, CZK_ERR_ARG, "czk_net_set_option: unknown name, value out of range, or slot_bytes after the first exchange");
    return CZK_OK;
}

extern "C" int czk_net_stats(const czk_net* n, uint64_t* out5) {
    if (!n || !out5) return CZK_ERR_ARG;
    memcpy(out5, n->stats, sizeof n->stats);
    return CZK_OK;
}
extern "C" void czk_net_stats_reset(czk_net* n) {
    if (n) memset(n->stats, 0, sizeof n->stats);
}

extern "C" int czk_net_broadcast(czk_net* n, const void* send, size_t bytes, void* recv, int mem) {
    CZK_TRY(exchange(n, Op::Broadcast, send, bytes, recv, mem));
    count_stats(n, Op::Broadcast, bytes);
    return CZK_OK;
}
extern "C" int czk_net_send_to_king(czk_net* n, const void* send, size_t bytes, void* recv, int mem) {
    CZK_TRY(exchange(n, Op::ToKing, send, bytes, recv, mem));
    count_stats(n, Op::ToKing, bytes);
    return CZK_OK;
}
extern "C" int czk_net_recv_from_king(czk_net* n, const void* send, size_t bytes, void* recv, int mem) {
    CZK_TRY(exchange(n, Op::FromKing, send, bytes, recv, mem));
    count_stats(n, Op::FromKing, bytes);
    return CZK_OK;
}
extern "C" int czk_net_barrier(czk_net* n) {
    if (!n) return CZK_ERR_ARG;
    if (n->shm_like()) {
        if (n->ctx) {
            NET_HIP(n, hipSetDevice(n->ctx->device));
            NET_HIP(n, hipStreamSynchronize(n->ctx->stream));
            n->reads_in_flight = false;
        }
        return shm_barrier(n);
    }
    uint8_t one = 1;
    std::vector<uint8_t> all(n->world);
    return exchange(n, Op::Broadcast, &one, 1, all.data(), CZK_MEM_HOST);
}

// ---- wire format -----------------------------------------------------------------------------------------------------------------------
extern "C" int czk_fr_vec_serialize(czk_ctx* ctx, const uint64_t* a, size_t n, int mem, uint8_t* out) {
    if (!ctx) return CZK_ERR_ARG;
    if (!out || (n && !a) || !valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "czk_fr_vec_serialize: null / bad argument");
    const uint64_t len = n;
    memcpy(out, &len, 8);   // u64 little-endian length (serialize/src/lib.rs:222-223); the hosts this library runs on are little-endian
    if (!n) return CZK_OK;
    if (mem == CZK_MEM_HOST) {
        std::vector<uint64_t> rep(4 * n);
        CZK_TRY(czk_fr_into_repr(ctx, a, rep.data(), n, CZK_MEM_HOST));
        memcpy(out + 8, rep.data(), 32 * n);
        return CZK_OK;
    }
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    DeviceBuf b;
    CZK_TRY(stage_take(ctx, 32 * n, &b));
    int rc = czk_fr_into_repr(ctx, a, (uint64_t*)b.p, n, CZK_MEM_DEVICE);
    if (rc == CZK_OK) rc = download_pageable(ctx, out + 8, b.p, 32 * n);   // blocking
    stage_give(ctx, b);
    return rc;
}

extern "C" int czk_fr_vec_deserialize(czk_ctx* ctx, const uint8_t* bytes, size_t len, uint64_t* out, size_t cap, int mem, size_t* n_out) {
    if (!ctx) return CZK_ERR_ARG;
    if (!bytes || !n_out || len < 8 || !valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "czk_fr_vec_deserialize: null / short input");
    uint64_t n64;
    memcpy(&n64, bytes, 8);
    if (n64 > (len - 8) / 32 || 8 + 32 * (size_t)n64 != len) return set_err(ctx, CZK_ERR_ARG, "Vec<Fr> wire format: length prefix does not match the payload");
    const size_t n = (size_t)n64;
    *n_out = n;
    if (n > cap || (n && !out)) return set_err(ctx, CZK_ERR_ARG, "czk_fr_vec_deserialize: more elements than the output holds");
    if (!n) return CZK_OK;
    std::vector<uint64_t> rep(4 * n);
    memcpy(rep.data(), bytes + 8, 32 * n);
    // Fp::deserialize -> from_repr fails on a non-canonical value (fields/macros.rs:443-454: `if r.is_valid()`)
    static const uint64_t R_MOD[4] = {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};
    for (size_t i = 0; i < n; i++) {
        bool lt = false;
        for (int j = 3; j >= 0; j--) {
            if (rep[4 * i + j] != R_MOD[j]) {
                lt = rep[4 * i + j] < R_MOD[j];
                break;
            }
        }
        if (!lt) return set_err(ctx, CZK_ERR_ARG, "Vec<Fr> wire format: element not below the modulus");
    }
    if (mem == CZK_MEM_HOST) return czk_fr_from_repr(ctx, rep.data(), out, n, CZK_MEM_HOST);
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    CZK_TRY(upload_pageable(ctx, out, rep.data(), 32 * n));   // returns once `rep` has been read
    return czk_fr_from_repr(ctx, out, out, n, CZK_MEM_DEVICE);
}

// ---- atomic_broadcast and the batch opens ----------------------------------------------------------------------------------------------
extern "C" int czk_net_atomic_broadcast(czk_net* n, const uint64_t* x, size_t cnt, uint64_t* recv, const uint8_t* rand32, int mem) {
    if (!n) return CZK_ERR_ARG;
    if (!n->ctx) return net_err(n, CZK_ERR_ARG, "czk_net_atomic_broadcast needs a communicator created with a context");
    if (cnt && (!x || !recv)) return net_err(n, CZK_ERR_ARG, "czk_net_atomic_broadcast: null buffer");
    const int W = n->world;
    const size_t ser = 8 + 32 * cnt;
    std::vector<uint8_t> wire(ser + 32), commit(32), all_commits(32 * (size_t)W), all_rnd(32 * (size_t)W);
    NET_CTX(n, czk_fr_vec_serialize(n->ctx, x, cnt, mem, wire.data()));
    if (rand32) memcpy(&wire[ser], rand32, 32);
    else if (getrandom(&wire[ser], 32, 0) != 32) return net_err(n, CZK_ERR_NET, "getrandom failed");
    czk_sha256(wire.data(), wire.size(), commit.data());                                          // commitment = H(data || randomness)
    CZK_TRY(exchange(n, Op::Broadcast, commit.data(), 32, all_commits.data(), CZK_MEM_HOST));   // exchange commitments
    CZK_TRY(exchange(n, Op::Broadcast, x, 32 * cnt, recv, mem));                                 // exchange (data || randomness)
    CZK_TRY(exchange(n, Op::Broadcast, &wire[ser], 32, all_rnd.data(), CZK_MEM_HOST));
    count_stats(n, Op::Broadcast, 32);              // the reference's two broadcasts: 32 bytes, then ser + 32 bytes (channel.rs:58-61)
    count_stats(n, Op::Broadcast, ser + 32);
    for (int p = 0; p < W; p++) {
        if (p == n->rank) continue;
        NET_CTX(n, czk_fr_vec_serialize(n->ctx, recv + 4 * cnt * (size_t)p, cnt, mem, wire.data()));
        memcpy(&wire[ser], &all_rnd[32 * (size_t)p], 32);
        uint8_t h[32];
        czk_sha256(wire.data(), wire.size(), h);
        if (memcmp(h, &all_commits[32 * (size_t)p], 32) != 0)
            return net_err(n, CZK_ERR_CHECK, "atomic_broadcast: party " + std::to_string(p) + "'s data does not match its commitment");
    }
    return CZK_OK;
}

namespace {
int open_args(czk_net* n, const void* a, const void* b, size_t cnt) {
    if (!n) return CZK_ERR_ARG;
    if (!n->ctx) return net_err(n, CZK_ERR_ARG, "batch opens need a communicator created with a context");
    if (cnt && (!a || !b)) return net_err(n, CZK_ERR_ARG, "batch open: null buffer");
    NET_HIP(n, hipSetDevice(n->ctx->device));
    return CZK_OK;
}
}  // namespace

extern "C" int czk_spdz_batch_open(czk_net* n, const uint64_t* sh, const uint64_t* mac, const uint64_t* mac_share, size_t cnt, uint64_t* out_value,
                                   int flags, uint64_t* out_bad) {
    CZK_TRY(open_args(n, sh, out_value, cnt));
    if (!out_bad || !mac_share || (cnt && !mac)) return net_err(n, CZK_ERR_ARG, "czk_spdz_batch_open: null argument");
    *out_bad = 0;
    if (!cnt) return CZK_OK;
    const size_t W = (size_t)n->world;
    CZK_TRY(net_buf(n, n->gather, W * cnt * 32));
    CZK_TRY(net_buf(n, n->dx, cnt * 32));
    uint64_t* g = (uint64_t*)n->gather.p;
    uint64_t* dx = (uint64_t*)n->dx.p;
    CZK_TRY(czk_net_broadcast(n, sh, cnt * 32, g, CZK_MEM_DEVICE));                       // let all_vals = Net::broadcast(&s_vals)
    NET_CTX(n, czk_fr_lanes_sum(n->ctx, g, W, cnt, out_value, nullptr));                  // vals[i] = sum over the parties
    NET_CTX(n, czk_fr_spdz_dx(n->ctx, out_value, mac, mac_share, dx, cnt));               // dx_t = mac_share * val - mac
    if (flags & CZK_OPEN_COMMIT) CZK_TRY(czk_net_atomic_broadcast(n, dx, cnt, g, nullptr, CZK_MEM_DEVICE));   // Net::atomic_broadcast(&dx_ts)
    else CZK_TRY(czk_net_broadcast(n, dx, cnt * 32, g, CZK_MEM_DEVICE));
    NET_CTX(n, czk_fr_lanes_sum(n->ctx, g, W, cnt, nullptr, out_bad));                    // assert!(sum.is_zero()) -- the caller's, on *out_bad
    return CZK_OK;
}

extern "C" int czk_add_batch_open(czk_net* n, const uint64_t* val, size_t cnt, uint64_t* out_value) {
    CZK_TRY(open_args(n, val, out_value, cnt));
    if (!cnt) return CZK_OK;
    const size_t W = (size_t)n->world;
    CZK_TRY(net_buf(n, n->gather, W * cnt * 32));
    CZK_TRY(czk_net_broadcast(n, val, cnt * 32, n->gather.p, CZK_MEM_DEVICE));
    NET_CTX(n, czk_fr_lanes_sum(n->ctx, (const uint64_t*)n->gather.p, W, cnt, out_value, nullptr));
    return CZK_OK;
}

extern "C" int czk_gsz_batch_open(czk_net* n, const uint64_t* val, size_t cnt, const uint32_t* degrees, unsigned degree, uint64_t* out_value,
                                  uint64_t* out_bad) {
    CZK_TRY(open_args(n, val, out_value, cnt));
    if (!out_bad) return net_err(n, CZK_ERR_ARG, "czk_gsz_batch_open: null out_bad");
    *out_bad = 0;
    if (!cnt) return CZK_OK;
    const size_t W = (size_t)n->world;
    CZK_TRY(net_buf(n, n->gather, W * cnt * 32));
    CZK_TRY(czk_net_broadcast(n, val, cnt * 32, n->gather.p, CZK_MEM_DEVICE));
    NET_CTX(n, czk_fr_gsz_open(n->ctx, (const uint64_t*)n->gather.p, W, cnt, degrees, degree, out_value, out_bad));
    return CZK_OK;
}

extern "C" int czk_fr_send_to_king(czk_net* n, const uint64_t* x, size_t cnt, uint64_t* gathered) {
    if (!n) return CZK_ERR_ARG;
    return czk_net_send_to_king(n, x, cnt * 32, gathered, CZK_MEM_DEVICE);
}
extern "C" int czk_fr_recv_from_king(czk_net* n, const uint64_t* parts, size_t cnt, uint64_t* out) {
    if (!n) return CZK_ERR_ARG;
    return czk_net_recv_from_king(n, parts, cnt * 32, out, CZK_MEM_DEVICE);
}

extern "C" int czk_gsz_batch_king_compute(czk_net* n, const uint64_t* val, size_t cnt, const uint32_t* degrees, unsigned degree, uint64_t* out,
                                          uint64_t* out_bad) {
    CZK_TRY(open_args(n, val, out, cnt));
    if (!out_bad) return net_err(n, CZK_ERR_ARG, "czk_gsz_batch_king_compute: null out_bad");
    *out_bad = 0;
    if (!cnt) return CZK_OK;
    const size_t W = (size_t)n->world;
    const bool king = n->rank == 0;
    // the king's scratch: world gathered lanes, then world identical answers (`vec![output; n]`, gsz20/mod.rs:508-512)
    if (king) CZK_TRY(net_buf(n, n->gather, 2 * W * cnt * 32));
    uint64_t* g = (uint64_t*)n->gather.p;
    CZK_TRY(czk_net_send_to_king(n, val, cnt * 32, king ? g : nullptr, CZK_MEM_DEVICE));
    uint64_t* ans = king ? g + 4 * W * cnt : nullptr;
    if (king) {
        NET_CTX(n, czk_fr_gsz_open(n->ctx, g, W, cnt, degrees, degree, ans, out_bad));   // open_degree_vec per element, f = identity
        for (size_t p = 1; p < W; p++)
            NET_HIP(n, hipMemcpyAsync(ans + 4 * cnt * p, ans, cnt * 32, hipMemcpyDeviceToDevice, n->ctx->stream));
    }
    return czk_net_recv_from_king(n, ans, cnt * 32, out, CZK_MEM_DEVICE);
}

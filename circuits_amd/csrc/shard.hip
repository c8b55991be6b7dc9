// One RollupMain batch sharded by transaction index over the ranks of a node, driven from ANY host language: hz_comm (a communicator:
// RCCL over xGMI, loaded with dlopen -- no link-time dependency --, or a host-staged transport over Unix sockets for boxes and tests
// without it) and hz_shard_step (the pass circuits_amd/multigpu.py's ShardedBatch.step makes from Python, inside the library).
// Reference counterpart: the `-n` thread-per-component mode of the compiled witness calculator (tools/helpers/actions.js:39-45) over the
// components src/rollup-main.circom:93-99 instantiates -- every DecodeTx / RollupTx is independent given the im* inputs.
//   rank r      transactions hz_shard_range(nTx, world, r): front / hash / SMT / signature kernels of its own range
//   collective 1   all_gather of the 160-byte data-availability records (the only thing HashInputs needs from other ranks)
//   rank 0      imports the records, FeeTx, the message, the sequential SHA-256 chain, the public output
//   collective 2   broadcast of the message blocks and chaining values (96 B per block)
//   rank r      the bit-level witness of its share of the blocks (hz_sha_expand)
// The witness stays sharded in HBM: rank r holds the signals of its transactions and blocks, rank 0 the fee section and the output.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <string>
#include <vector>
#include "hostutil.h"
#include "ctx_internal.h"

using namespace hz;

namespace {
// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------------------
struct Rccl {
    void* so = nullptr;
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (so) return true;
        const char* names[] = {getenv("HZ_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (so) break;
            err = dlerror();
        }
        if (!so) return false;
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(so, n); if (!p) { ok = false; err = std::string("librccl lacks ") + n; } return p; };
        GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
        CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
        CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
        Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
        GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        if (!ok) { dlclose(so); so = nullptr; }
        return ok;
    }
};
Rccl& rccl() { static Rccl r; return r; }
const int kNcclUint8 = 1;

// ---- the rendezvous / host-staged transport: a star of Unix stream sockets, rank 0 at the centre ----------------------------------
bool write_all(int fd, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    while (n) {
        const ssize_t w = ::send(fd, b, n, MSG_NOSIGNAL);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        b += w; n -= (size_t)w;
    }
    return true;
}
bool read_all(int fd, void* p, size_t n) {
    uint8_t* b = (uint8_t*)p;
    while (n) {
        const ssize_t r = ::recv(fd, b, n, 0);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;
        b += r; n -= (size_t)r;
    }
    return true;
}
}  // namespace

struct hz_comm {
    int transport = 0, device = 0, rank = 0, world = 1;
    void* nccl = nullptr;            // ncclComm_t
    int listen_fd = -1;
    std::vector<int> peer;           // rank 0: fd of rank r at [r]; rank r > 0: fd of rank 0 at [0]
    std::string path;
    std::vector<uint8_t> host;       // staging of the socket transport
    // what hz_shard_step keeps between passes (sized for the context it was first given)
    const hz_ctx* bound = nullptr;
    DevBuf send, recv, sha;
    std::vector<std::pair<int32_t, int32_t>> ranges;
    size_t slot = 0;
    ~hz_comm() {
        if (nccl && rccl().CommDestroy) (void)rccl().CommDestroy(nccl);
        for (int fd : peer) if (fd >= 0) ::close(fd);
        if (listen_fd >= 0) { ::close(listen_fd); if (!path.empty()) ::unlink(path.c_str()); }
    }
};

static hz_status comm_rendezvous(hz_comm* c, const char* path) {
    if (c->world == 1) return HZ_OK;
    if (!path || !*path || strlen(path) >= sizeof(((sockaddr_un*)nullptr)->sun_path))
        return set_err(HZ_ERR_ARG, "hz_comm_create: a rendezvous path (a Unix socket name shorter than %zu bytes) is needed for more than one rank", sizeof(((sockaddr_un*)nullptr)->sun_path));
    c->path = path;
    sockaddr_un sa;
    memset(&sa, 0, sizeof sa);
    sa.sun_family = AF_UNIX;
    strcpy(sa.sun_path, path);
    if (c->rank == 0) {
        c->listen_fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        if (c->listen_fd < 0) return set_err(HZ_ERR_ARG, "hz_comm_create: socket(): %s", strerror(errno));
        ::unlink(path);
        if (::bind(c->listen_fd, (sockaddr*)&sa, sizeof sa) != 0 || ::listen(c->listen_fd, c->world) != 0)
            return set_err(HZ_ERR_ARG, "hz_comm_create: cannot listen on %s: %s", path, strerror(errno));
        c->peer.assign((size_t)c->world, -1);
        for (int k = 1; k < c->world; k++) {
            const int fd = ::accept(c->listen_fd, nullptr, nullptr);
            int32_t r = -1;
            if (fd < 0 || !read_all(fd, &r, sizeof r) || r <= 0 || r >= c->world || c->peer[(size_t)r] >= 0) {
                if (fd >= 0) ::close(fd);
                return set_err(HZ_ERR_ARG, "hz_comm_create: bad rendezvous on %s (rank %d announced)", path, r);
            }
            c->peer[(size_t)r] = fd;
        }
    } else {
        const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) return set_err(HZ_ERR_ARG, "hz_comm_create: socket(): %s", strerror(errno));
        // rank 0 may not be listening yet: retry for HZ_COMM_TIMEOUT_S seconds (default 120)
        const int limit = getenv("HZ_COMM_TIMEOUT_S") ? atoi(getenv("HZ_COMM_TIMEOUT_S")) : 120;
        bool ok = false;
        for (int tries = 0; tries < limit * 20 && !ok; tries++) {
            ok = ::connect(fd, (sockaddr*)&sa, sizeof sa) == 0;
            if (!ok) { timespec ts{0, 50 * 1000 * 1000}; nanosleep(&ts, nullptr); }
        }
        const int32_t r = c->rank;
        if (!ok || !write_all(fd, &r, sizeof r)) { ::close(fd); return set_err(HZ_ERR_ARG, "hz_comm_create: rank %d cannot reach rank 0 at %s", c->rank, path); }
        c->peer.assign(1, fd);
    }
    return HZ_OK;
}
// rank 0's `n` bytes to everybody (host memory)
static bool host_bcast(hz_comm* c, void* buf, size_t n) {
    if (c->world == 1) return true;
    if (c->rank == 0) { for (int r = 1; r < c->world; r++) if (!write_all(c->peer[(size_t)r], buf, n)) return false; return true; }
    return read_all(c->peer[0], buf, n);
}

extern "C" hz_status hz_comm_create(int32_t transport, int32_t device, int32_t rank, int32_t world, const char* rendezvous_path, hz_comm** out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (transport != HZ_COMM_RCCL && transport != HZ_COMM_SOCKET))
        return set_err(HZ_ERR_ARG, "hz_comm_create: bad argument (transport %d, rank %d of %d)", transport, rank, world);
    if (hz_device_count() <= 0) return set_err(HZ_ERR_NODEVICE, "no usable gfx950 device");
    if (device < 0 || device >= hz_device_count()) return set_err(HZ_ERR_ARG, "hz_comm_create: bad device ordinal %d", device);
    hz_comm* c = new hz_comm();
    c->transport = transport; c->device = device; c->rank = rank; c->world = world;
    hz_status st = comm_rendezvous(c, rendezvous_path);
    if (st == HZ_OK && transport == HZ_COMM_RCCL) {
        Rccl& R = rccl();
        if (!R.load()) st = set_err(HZ_ERR_ARG, "hz_comm_create: RCCL is not loadable (%s); HZ_RCCL_LIB names the library", R.err.c_str());
        Rccl::UniqueId id;
        memset(&id, 0, sizeof id);
        if (st == HZ_OK && rank == 0) { const int e = R.GetUniqueId(&id); if (e) st = set_err(HZ_ERR_ARG, "ncclGetUniqueId: %s", R.GetErrorString(e)); }
        if (st == HZ_OK && !host_bcast(c, &id, sizeof id)) st = set_err(HZ_ERR_ARG, "hz_comm_create: the RCCL id did not reach every rank over %s", rendezvous_path ? rendezvous_path : "?");
        if (st == HZ_OK) {
            if (hipSetDevice(device) != hipSuccess) st = set_err(HZ_ERR_HIP, "hz_comm_create: hipSetDevice(%d)", device);
            else { const int e = R.CommInitRank(&c->nccl, world, id, rank); if (e) st = set_err(HZ_ERR_ARG, "ncclCommInitRank (rank %d of %d): %s", rank, world, R.GetErrorString(e)); }
        }
    }
    if (st != HZ_OK) { delete c; return st; }
    *out = c;
    return HZ_OK;
}
extern "C" void hz_comm_destroy(hz_comm* c) { delete c; }
extern "C" int32_t hz_comm_rank(const hz_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t hz_comm_world(const hz_comm* c) { return c ? c->world : 0; }

// every rank's `n` bytes at d_send into d_recv[r * n] of every rank, ordered on `s`
static hz_status comm_all_gather(hz_comm* c, const void* d_send, void* d_recv, size_t n, hipStream_t s) {
    if (c->transport == HZ_COMM_RCCL) {
        const int e = rccl().AllGather(d_send, d_recv, n, kNcclUint8, c->nccl, s);
        return e ? set_err(HZ_ERR_ARG, "ncclAllGather: %s", rccl().GetErrorString(e)) : HZ_OK;
    }
    // host-staged: everybody's block to rank 0, the assembled buffer back (latency-sized: 47-330 KB per pass)
    c->host.resize(n * (size_t)c->world);
    uint8_t* h = c->host.data();
    HZ_HIP(hipMemcpyAsync(h + (size_t)c->rank * n, d_send, n, hipMemcpyDeviceToHost, s));
    HZ_HIP(hipStreamSynchronize(s));
    bool ok = true;
    if (c->rank == 0) {
        for (int r = 1; r < c->world && ok; r++) ok = read_all(c->peer[(size_t)r], h + (size_t)r * n, n);
    } else ok = write_all(c->peer[0], h + (size_t)c->rank * n, n);
    ok = ok && host_bcast(c, h, n * (size_t)c->world);
    if (!ok) return set_err(HZ_ERR_ARG, "hz_shard_step: the all_gather over %s failed on rank %d (%s)", c->path.c_str(), c->rank, strerror(errno));
    HZ_HIP(hipMemcpyAsync(d_recv, h, n * (size_t)c->world, hipMemcpyHostToDevice, s));
    HZ_HIP(hipStreamSynchronize(s));   // (the staging vector is reused by the next collective)
    return HZ_OK;
}
static hz_status comm_broadcast(hz_comm* c, void* d_buf, size_t n, hipStream_t s) {
    if (c->world == 1 && c->transport != HZ_COMM_RCCL) return HZ_OK;
    if (c->transport == HZ_COMM_RCCL) {
        const int e = rccl().Broadcast(d_buf, d_buf, n, kNcclUint8, 0, c->nccl, s);
        return e ? set_err(HZ_ERR_ARG, "ncclBroadcast: %s", rccl().GetErrorString(e)) : HZ_OK;
    }
    c->host.resize(n);
    if (c->rank == 0) { HZ_HIP(hipMemcpyAsync(c->host.data(), d_buf, n, hipMemcpyDeviceToHost, s)); HZ_HIP(hipStreamSynchronize(s)); }
    if (!host_bcast(c, c->host.data(), n)) return set_err(HZ_ERR_ARG, "hz_shard_step: the broadcast over %s failed on rank %d", c->path.c_str(), c->rank);
    if (c->rank != 0) { HZ_HIP(hipMemcpyAsync(d_buf, c->host.data(), n, hipMemcpyHostToDevice, s)); HZ_HIP(hipStreamSynchronize(s)); }
    return HZ_OK;
}

extern "C" hz_status hz_shard_step(hz_ctx* ctx, hz_comm* c, void* stream) {
    if (!ctx || !c) return set_err(HZ_ERR_ARG, "hz_shard_step: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (!s) {   // NULL: the context's own stream, as in every entry point -- the collectives are ordered with the export / import kernels on it
        CtxGeom g;
        ctx_geometry(ctx, g);
        s = g.s_main;
    }
    HZ_HIP(hipSetDevice(c->device));
    if (c->bound != ctx) {   // first pass with this context: ranges, exchange buffers, the shard itself
        const uint64_t blocks = hz_sha_blocks(ctx);
        if (!blocks) return set_err(HZ_ERR_ARG, "hz_shard_step: RollupMain contexts only");
        // nTx: the largest range starts where the last rank's ends
        int32_t f = 0, n = 0;
        c->ranges.clear();
        // hz_ctx_set_shard validates against the context's nTx; ask the library for it through the record count of the whole batch
        const int32_t nTx = (int32_t)hz_ctx_ntx(ctx);
        size_t mx = 0;
        for (int r = 0; r < c->world; r++) { hz_shard_range(nTx, c->world, r, &f, &n); c->ranges.push_back({f, n}); mx = std::max(mx, (size_t)n); }
        c->slot = std::max<size_t>(mx, 1) * hz_da_record_bytes(ctx);
        HZ_HIP(c->send.alloc(c->slot));
        HZ_HIP(c->recv.alloc(c->slot * (size_t)c->world));
        HZ_HIP(c->sha.alloc(hz_sha_state_bytes(ctx)));
        HZ_HIP(hipMemset(c->send.p, 0, c->slot));
        HZ_HIP(hipMemset(c->recv.p, 0, c->slot * (size_t)c->world));
        const hz_status st = hz_ctx_set_shard(ctx, c->ranges[(size_t)c->rank].first, c->ranges[(size_t)c->rank].second, c->rank == 0);
        if (st != HZ_OK) return st;
        c->bound = ctx;
    }
    hz_status st = hz_witness_enqueue(ctx, s);                        // this rank's transactions
    if (st == HZ_OK) st = hz_da_export(ctx, c->send.p, s);
    if (st == HZ_OK) st = comm_all_gather(c, c->send.p, c->recv.p, c->slot, s);   // collective 1: the data-availability records
    if (st != HZ_OK) return st;
    if (c->rank == 0) {
        for (int r = 1; r < c->world && st == HZ_OK; r++)
            st = hz_da_import(ctx, c->ranges[(size_t)r].first, c->ranges[(size_t)r].second, (const uint8_t*)c->recv.p + (size_t)r * c->slot, s);
        if (st == HZ_OK) st = hz_witness_enqueue_tail_chain(ctx, s);  // FeeTx + message + the sequential SHA-256 chain
        if (st == HZ_OK) st = hz_sha_export(ctx, c->sha.p, s);
        if (st != HZ_OK) return st;
    }
    st = comm_broadcast(c, c->sha.p, c->sha.bytes, s);                // collective 2: message blocks and chaining values from rank 0
    if (st != HZ_OK) return st;
    int32_t bf = 0, bn = 0;
    hz_shard_range((int32_t)hz_sha_blocks(ctx), c->world, c->rank, &bf, &bn);
    return hz_sha_expand(ctx, bf, bn, c->rank == 0 ? nullptr : c->sha.p, s);   // then hz_witness_check on every rank
}

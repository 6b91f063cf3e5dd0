// fake_rccl.cpp -- TEST DOUBLE for librccl, used by tests/ and by bench.py's functional N > 1 lines on a one-GPU box only (never by
// the product: libgnnagg.so loads the real librccl unless the test hook GNNAGG_RCCL_LIB names another library).
//
// RCCL needs one GPU per rank, the test box has one GPU.  This library implements the eight entry points dist_rccl.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString) for
// ranks that are PROCESSES SHARING ONE GPU.
//
// Round 6: the double is STREAM-ORDERED AND ASYNCHRONOUS, like the real thing.  ncclGroupEnd only ENQUEUES work on
// the caller's stream and returns; nothing in it waits for the GPU (no hipStreamSynchronize, no hipMemcpy through the host):
//
//   * a message src -> dst is two device-to-device hipMemcpyAsync: on the SENDER's stream out of the user's buffer into a staging arena
//     the double owns (one per direction, grown on demand), on the RECEIVER's stream out of that arena -- mapped there through IPC
//     (hipIpcGetMemHandle / hipIpcOpenMemHandle, once per arena) -- into the user's receive buffer.  (The arenas, not the users'
//     allocations, are exported: on this ROCm hipIpcGetMemHandle returns a different handle on every call, so exports must be cached,
//     and a cache keyed by a user's address goes stale when that address is freed and reused -- selftest.cpp prints both facts.);
//   * the order between the two processes' streams is kept on the DEVICE by two counters per direction that live in IPC-shared
//     device memory: `ready` (bumped by a one-thread kernel on the sender's stream behind its staging copies: everything enqueued
//     before the group has run, the message is in place) and `done` (bumped on the receiver's stream behind its copy: the arena may be
//     rewritten).  The receiver's stream polls `ready` in a tiny kernel in front of its copy, the sender's stream polls `done` at the
//     end of its group -- like RCCL's fused send / recv kernel, a group is over on a stream when its peers have taken its messages;
//   * the counters' targets are kept on the device as well (every wait kernel advances its own `want` word), so a group's stream
//     operations carry no sequence numbers and a CAPTURED step replays correctly (all ranks replay the same number of times);
//   * the hosts only exchange a 128-byte descriptor per message (IPC handle, offset, byte count: checked against the receiver's) through
//     a ring in /dev/shm -- a host-to-host rendezvous of the ENQUEUE, as RCCL's proxies have one, never a wait for device work.
//
// So the fork / join / event ordering of the step code in dist_rccl.cpp is exercised in the regime it is written for: when
// gnnagg_dist_step_* returns, the halo rows are NOT there yet; only the events order the halo-source passes behind them.  Test knobs:
// FAKE_RCCL_DELAY_US (+ FAKE_RCCL_DELAY_RANKS = comma list, default all): a spin kernel of that length in front of every group of those
// ranks -- a peer whose data arrives late; FAKE_RCCL_TIMEOUT_S (default 60): a device-side wait that long gives up, sets an error the
// next nccl call reports, and the test fails on its data instead of hanging the GPU.  What it keeps of the real thing is the contract
// the C-ABI step relies on: point-to-point messages matched per (source, destination) in posting order, byte counts that must agree on
// both ends, one stream per group, completion in stream order.  All ranks must sit on the same physical device (checked by PCI bus id).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int kMaxPeers = 64;
constexpr int kRing = 64;            // descriptors a sender may post ahead of the receiver's enqueue
constexpr double kHostTimeoutS = 120.0;

struct Desc {                        // one message, as the sender describes it to the receiver's host
    hipIpcMemHandle_t handle;        // the sender's staging arena for this direction
    uint64_t alloc_bytes, offset, bytes;
    uint64_t pad[5];
};
static_assert(sizeof(Desc) == 128, "descriptor size");

struct BoxHeader {                   // one direction of one pair: file /dev/shm/fakerccl_<token>_<src>_<dst>, created by src
    std::atomic<uint64_t> posted;    // descriptors the sender has published
    std::atomic<uint64_t> taken;     // descriptors the receiver has read
    uint64_t pad[6];
    Desc ring[kRing];
};

struct RankHeader {                  // file /dev/shm/fakerccl_<token>_rank<r>
    hipIpcMemHandle_t flags;         // the rank's device counters
    char bus[64];                    // PCI bus id of its device
    std::atomic<uint32_t> leaving;   // set in ncclCommDestroy after the device has drained
};

struct Flags {                       // device memory of one rank, mapped by every peer
    unsigned ready[kMaxPeers];       // ready[p]: messages p -> me whose data is in place       (written by p)
    unsigned done[kMaxPeers];        // done[p]:  messages me -> p that p has copied out         (written by p)
    unsigned want_ready[kMaxPeers];  // the next values this rank's wait kernels expect (local, advanced by the kernels themselves)
    unsigned want_done[kMaxPeers];
};

struct WordList {                    // kernel argument: up to one word per peer
    unsigned *word[kMaxPeers];
    unsigned *want[kMaxPeers];
    unsigned add[kMaxPeers];
    int n;
};

__global__ void k_signal(WordList l)
{
    const int i = threadIdx.x;
    if (i < l.n) __hip_atomic_fetch_add(l.word[i], l.add[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_wait(WordList l, unsigned *err, long long timeout_ticks)
{
    const int i = threadIdx.x;
    if (i < l.n) {
        const unsigned w = *l.want[i] + l.add[i];
        *l.want[i] = w;   // (waits on one word are stream-ordered with each other: only they touch `want`)
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(l.word[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - w) < 0) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(err, 1u + (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

__global__ void k_delay(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

struct Mapped {
    int fd = -1;
    void *map = nullptr;
    size_t bytes = 0;
    bool open_file(const std::string &path, size_t n, bool create)
    {
        bytes = n;
        if (create) {
            const std::string tmp = path + ".tmp";
            fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)n) != 0) return false;
            map = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            return map != MAP_FAILED;   // (published by the caller with rename once it is filled in)
        }
        fd = open(path.c_str(), O_RDWR);
        if (fd < 0) return false;
        map = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        return map != MAP_FAILED;
    }
    void close_file()
    {
        if (map && map != MAP_FAILED) munmap(map, bytes);
        if (fd >= 0) close(fd);
        map = nullptr; fd = -1;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_ops;

struct Arena { char *base = nullptr; size_t cap = 0; hipIpcMemHandle_t handle; };
}  // namespace

struct ncclComm {
    int rank = 0, world = 1;
    std::string token;
    Mapped self_file;                       // RankHeader of this rank
    std::vector<Mapped> peer_file;          // RankHeader of every rank (self: unused)
    std::vector<Mapped> out, in;            // out[p]: BoxHeader rank -> p (created here), in[p]: p -> rank
    std::vector<uint64_t> n_out, n_in;      // descriptors posted / taken per peer
    Flags *flags = nullptr;                 // this rank's counters (device)
    std::vector<Flags *> peer_flags;        // every rank's counters as mapped here (self: flags)
    unsigned *err = nullptr;                // pinned host word a timed-out wait kernel sets
    long long timeout_ticks = 0, delay_ticks = 0;
    std::vector<Arena> arena;               // arena[p]: staging of the messages rank -> p (exported once per allocation)
    std::vector<void *> retired;            // outgrown arenas: a peer may still be reading them, freed with the communicator
    std::vector<std::map<std::string, void *>> opened;   // per peer: handle bytes -> mapping of its arena
    std::string path(int src, int dst) const { return "/dev/shm/fakerccl_" + token + "_" + std::to_string(src) + "_" + std::to_string(dst); }
    std::string rank_path(int r) const { return "/dev/shm/fakerccl_" + token + "_rank" + std::to_string(r); }
    BoxHeader *box_out(int p) { return static_cast<BoxHeader *>(out[p].map); }
    BoxHeader *box_in(int p) { return static_cast<BoxHeader *>(in[p].map); }
};

static void release_comm(ncclComm *c)
{
    for (auto &m : c->opened)
        for (auto &e : m) (void)hipIpcCloseMemHandle(e.second);
    for (int p = 0; p < (int)c->peer_flags.size(); ++p)
        if (p != c->rank && c->peer_flags[p]) (void)hipIpcCloseMemHandle(c->peer_flags[p]);
    for (Arena &a : c->arena) if (a.base) (void)hipFree(a.base);
    for (void *p : c->retired) (void)hipFree(p);
    if (c->flags) (void)hipFree(c->flags);
    if (c->err) (void)hipHostFree(c->err);
    for (int p = 0; p < (int)c->out.size(); ++p) {
        c->out[p].close_file();
        (void)unlink(c->path(c->rank, p).c_str());
    }
    for (auto &m : c->in) m.close_file();
    for (auto &m : c->peer_file) m.close_file();
    c->self_file.close_file();
    (void)unlink(c->rank_path(c->rank).c_str());
    delete c;
}

// room for `need` bytes in the arena towards `peer`; a larger arena replaces an outgrown one (never inside a steady-state step)
static ncclResult_t reserve_arena(ncclComm *c, int peer, size_t need)
{
    Arena &a = c->arena[(size_t)peer];
    if (need <= a.cap) return ncclSuccess;
    size_t cap = a.cap ? a.cap * 2 : (size_t)1 << 20;
    while (cap < need) cap *= 2;
    if (a.base) c->retired.push_back(a.base);
    a.base = nullptr; a.cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&a.base), cap);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&a.handle, a.base);
    if (e != hipSuccess) {
        fprintf(stderr, "fake_rccl: rank %d: staging arena of %zu bytes towards rank %d: %s\n", c->rank, cap, peer, hipGetErrorString(e));
        return ncclUnhandledCudaError;
    }
    a.cap = cap;
    return ncclSuccess;
}

static ncclResult_t flush_comm(ncclComm *c, const std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    if (__atomic_load_n(c->err, __ATOMIC_RELAXED) != 0) {
        fprintf(stderr, "fake_rccl: rank %d: a device-side wait for peer %u timed out earlier\n", c->rank, *c->err - 1);
        return ncclSystemError;
    }
    hipStream_t stream = ops[0].stream;
    for (const Op &o : ops)
        if (o.stream != stream) return ncclInvalidUsage;   // (one stream per group is all the step code uses)
    if (c->delay_ticks > 0) hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, stream, c->delay_ticks);
    // messages to itself: matched inside the group, a stream-ordered copy
    {
        std::vector<const Op *> s, r;
        for (const Op &o : ops)
            if (o.peer == c->rank) (o.send ? s : r).push_back(&o);
        if (s.size() != r.size()) return ncclInvalidUsage;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i]->bytes != r[i]->bytes) return ncclInvalidArgument;
            if (s[i]->bytes && hipMemcpyAsync(r[i]->buf, s[i]->buf, s[i]->bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    // 1. sends: stage every message in this direction's arena (a copy on this stream), describe it to the receiver's host, then ONE
    // kernel says "in place" to all receivers
    WordList sig{}, fin{};
    std::vector<int> slot_of_peer((size_t)c->world, -1);
    std::vector<size_t> need((size_t)c->world, 0), used((size_t)c->world, 0);
    auto padded = [](size_t n) { return (n + 255) & ~(size_t)255; };
    for (const Op &o : ops)
        if (o.send && o.peer != c->rank) need[(size_t)o.peer] += padded(o.bytes);
    for (int p = 0; p < c->world; ++p)
        if (need[(size_t)p]) { const ncclResult_t r = reserve_arena(c, p, need[(size_t)p]); if (r != ncclSuccess) return r; }
    for (const Op &o : ops) {
        if (!o.send || o.peer == c->rank) continue;
        BoxHeader *b = c->box_out(o.peer);
        const double t0 = now_s();
        while (c->n_out[o.peer] - b->taken.load(std::memory_order_acquire) >= (uint64_t)kRing) {
            if (now_s() - t0 > kHostTimeoutS) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        const Arena &a = c->arena[(size_t)o.peer];
        Desc d{};
        d.bytes = o.bytes;
        d.handle = a.handle;
        d.alloc_bytes = a.cap;
        d.offset = used[(size_t)o.peer];
        used[(size_t)o.peer] += padded(o.bytes);
        if (o.bytes && hipMemcpyAsync(a.base + d.offset, o.buf, o.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        b->ring[c->n_out[o.peer] % kRing] = d;
        b->posted.store(++c->n_out[o.peer], std::memory_order_release);
        int &slot = slot_of_peer[(size_t)o.peer];
        if (slot < 0) {
            slot = sig.n++;
            sig.word[slot] = &c->peer_flags[o.peer]->ready[c->rank];
            fin.word[slot] = &c->flags->done[o.peer];
            fin.want[slot] = &c->flags->want_done[o.peer];
            fin.n = sig.n;
        }
        sig.add[slot] += 1;
        fin.add[slot] += 1;
    }
    if (sig.n) hipLaunchKernelGGL(k_signal, dim3(1), dim3(kMaxPeers), 0, stream, sig);
    // 2. receives, in posting order: wait for the sender's "in place", copy out of its buffer, tell it "copied"
    for (const Op &o : ops) {
        if (o.send || o.peer == c->rank) continue;
        BoxHeader *b = c->box_in(o.peer);
        const uint64_t want = c->n_in[o.peer] + 1;
        const double t0 = now_s();
        while (b->posted.load(std::memory_order_acquire) < want) {
            if (now_s() - t0 > kHostTimeoutS) {
                fprintf(stderr, "fake_rccl: rank %d: rank %d did not post its send in time\n", c->rank, o.peer);
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        const Desc d = b->ring[(want - 1) % kRing];
        c->n_in[o.peer] = want;
        b->taken.store(want, std::memory_order_release);
        if (d.bytes != o.bytes) {
            fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)d.bytes);
            return ncclInvalidArgument;
        }
        const char *src = nullptr;
        if (o.bytes) {
            const std::string key(reinterpret_cast<const char *>(&d.handle), sizeof(d.handle));
            auto &cache = c->opened[(size_t)o.peer];
            auto it = cache.find(key);
            if (it == cache.end()) {
                void *m = nullptr;
                const hipError_t e = hipIpcOpenMemHandle(&m, d.handle, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess || !m) {
                    fprintf(stderr, "fake_rccl: rank %d: hipIpcOpenMemHandle (rank %d's staging arena, %llu bytes): %s\n", c->rank, o.peer, (unsigned long long)d.alloc_bytes, hipGetErrorString(e));
                    return ncclUnhandledCudaError;
                }
                it = cache.emplace(key, m).first;
            }
            src = static_cast<const char *>(it->second) + d.offset;
        }
        WordList w{};
        w.n = 1;
        w.word[0] = &c->flags->ready[o.peer];
        w.want[0] = &c->flags->want_ready[o.peer];
        w.add[0] = 1;
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(kMaxPeers), 0, stream, w, c->err, c->timeout_ticks);
        if (o.bytes && hipMemcpyAsync(o.buf, src, o.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
        WordList s{};
        s.n = 1;
        s.word[0] = &c->peer_flags[o.peer]->done[c->rank];
        s.add[0] = 1;
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(kMaxPeers), 0, stream, s);
    }
    // 3. the group is over on this stream when every receiver has copied (the arenas may be rewritten by the next group)
    if (fin.n) hipLaunchKernelGGL(k_wait, dim3(1), dim3(kMaxPeers), 0, stream, fin, c->err, c->timeout_ticks);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

static ncclResult_t flush_ops()
{
    std::vector<std::pair<ncclComm *, Op>> ops;
    ops.swap(g_ops);
    std::vector<ncclComm *> comms;
    for (auto &e : ops) {
        bool seen = false;
        for (ncclComm *c : comms) seen = seen || c == e.first;
        if (!seen) comms.push_back(e.first);
    }
    for (ncclComm *c : comms) {
        std::vector<Op> mine;
        for (auto &e : ops)
            if (e.first == c) mine.push_back(e.second);
        const ncclResult_t r = flush_comm(c, mine);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    std::random_device rd;
    snprintf(id->internal, sizeof(id->internal), "%08x%08x%08x", rd(), rd(), (unsigned)getpid());
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > kMaxPeers || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->rank = rank; c->world = nranks;
    c->token.assign(id.internal, strnlen(id.internal, sizeof(id.internal)));
    c->out.resize(nranks); c->in.resize(nranks); c->peer_file.resize(nranks);
    c->n_out.assign(nranks, 0); c->n_in.assign(nranks, 0);
    c->peer_flags.assign(nranks, nullptr);
    c->opened.resize(nranks);
    c->arena.resize(nranks);
    int dev = 0, khz = 100000;
    if (hipGetDevice(&dev) != hipSuccess) { release_comm(c); return ncclUnhandledCudaError; }
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    if (khz <= 0) khz = 100000;
    const char *e;
    const double timeout_s = (e = getenv("FAKE_RCCL_TIMEOUT_S")) ? atof(e) : 60.0;
    c->timeout_ticks = (long long)(timeout_s * 1e3 * khz);
    if ((e = getenv("FAKE_RCCL_DELAY_US")) && atof(e) > 0) {
        bool mine = true;
        if (const char *r = getenv("FAKE_RCCL_DELAY_RANKS")) {
            mine = false;
            for (const char *p = r; *p;) {
                char *end;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                mine = mine || v == rank;
                p = *end ? end + 1 : end;
            }
        }
        if (mine) c->delay_ticks = (long long)(atof(e) * 1e-3 * khz);
    }
    if (hipMalloc(reinterpret_cast<void **>(&c->flags), sizeof(Flags)) != hipSuccess || hipMemset(c->flags, 0, sizeof(Flags)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess || hipHostMalloc(reinterpret_cast<void **>(&c->err), sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
        release_comm(c);
        return ncclUnhandledCudaError;
    }
    *c->err = 0;
    c->peer_flags[rank] = c->flags;
    // what this rank publishes: its header (counter handle, bus id) and its outgoing boxes -- filled in, then renamed into place
    if (!c->self_file.open_file(c->rank_path(rank), sizeof(RankHeader), true)) { release_comm(c); return ncclSystemError; }
    RankHeader *me = static_cast<RankHeader *>(c->self_file.map);
    memset(me, 0, sizeof(*me));
    if (nranks > 1 && hipIpcGetMemHandle(&me->flags, c->flags) != hipSuccess) {
        fprintf(stderr, "fake_rccl: rank %d: hipIpcGetMemHandle of the counters failed: %s\n", rank, hipGetErrorString(hipGetLastError()));
        release_comm(c);
        return ncclUnhandledCudaError;
    }
    (void)hipDeviceGetPCIBusId(me->bus, sizeof(me->bus), dev);
    if (rename((c->rank_path(rank) + ".tmp").c_str(), c->rank_path(rank).c_str()) != 0) { release_comm(c); return ncclSystemError; }
    for (int p = 0; p < nranks; ++p) {
        if (!c->out[p].open_file(c->path(rank, p), sizeof(BoxHeader), true)) { release_comm(c); return ncclSystemError; }
        memset(c->out[p].map, 0, sizeof(BoxHeader));
        if (rename((c->path(rank, p) + ".tmp").c_str(), c->path(rank, p).c_str()) != 0) { release_comm(c); return ncclSystemError; }
    }
    const double t0 = now_s();
    for (int p = 0; p < nranks; ++p) {   // what the peers publish
        while (!c->in[p].open_file(c->path(p, rank), sizeof(BoxHeader), false) ) {
            c->in[p].close_file();
            if (now_s() - t0 > kHostTimeoutS) { release_comm(c); return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (p == rank) continue;
        while (!c->peer_file[p].open_file(c->rank_path(p), sizeof(RankHeader), false)) {
            c->peer_file[p].close_file();
            if (now_s() - t0 > kHostTimeoutS) { release_comm(c); return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        RankHeader *h = static_cast<RankHeader *>(c->peer_file[p].map);
        if (strncmp(h->bus, me->bus, sizeof(me->bus)) != 0) {
            fprintf(stderr, "fake_rccl: rank %d is on device %s, rank %d on %s: this double is for ranks SHARING one GPU\n", rank, me->bus, p, h->bus);
            release_comm(c);
            return ncclInvalidUsage;
        }
        void *m = nullptr;
        const hipError_t err = hipIpcOpenMemHandle(&m, h->flags, hipIpcMemLazyEnablePeerAccess);
        if (err != hipSuccess || !m) {
            fprintf(stderr, "fake_rccl: rank %d: hipIpcOpenMemHandle of rank %d's counters: %s\n", rank, p, hipGetErrorString(err));
            release_comm(c);
            return ncclUnhandledCudaError;
        }
        c->peer_flags[p] = static_cast<Flags *>(m);
    }
    *comm = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    // the peers' kernels write this rank's counters and read its buffers: leave together, after every device has drained
    (void)hipDeviceSynchronize();
    if (*comm->err) fprintf(stderr, "fake_rccl: rank %d: a device-side wait for peer %u timed out\n", comm->rank, *comm->err - 1);
    static_cast<RankHeader *>(comm->self_file.map)->leaving.store(1, std::memory_order_release);
    const double t0 = now_s();
    for (int p = 0; p < comm->world; ++p) {
        if (p == comm->rank) continue;
        RankHeader *h = static_cast<RankHeader *>(comm->peer_file[p].map);
        while (h->leaving.load(std::memory_order_acquire) == 0 && now_s() - t0 < 20.0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    release_comm(comm);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush_ops();
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->world || datatype != ncclInt8) return ncclInvalidArgument;   // (the C-ABI moves bytes)
    g_ops.push_back({comm, Op{true, const_cast<void *>(sendbuff), count, peer, stream}});
    return g_depth > 0 ? ncclSuccess : flush_ops();
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->world || datatype != ncclInt8) return ncclInvalidArgument;
    g_ops.push_back({comm, Op{false, recvbuff, count, peer, stream}});
    return g_depth > 0 ? ncclSuccess : flush_ops();
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "fake_rccl: success";
        case ncclSystemError: return "fake_rccl: a peer did not answer in time (host rendezvous or device-side wait)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument or mismatched message size";
        case ncclInvalidUsage: return "fake_rccl: invalid usage (one stream per group; ranks must share one GPU)";
        case ncclUnhandledCudaError: return "fake_rccl: HIP error";
        default: return "fake_rccl: error";
    }
}

// test-only knob beside the environment variables: the delay in front of the calling rank's following groups (0: none)
__attribute__((visibility("default"))) void fakeRcclSetDelayUs(ncclComm_t comm, double us)
{
    int khz = 100000, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    if (comm) comm->delay_ticks = (long long)(us * 1e-3 * (khz > 0 ? khz : 100000));
}

}  // extern "C"

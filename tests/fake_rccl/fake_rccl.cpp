// fake_rccl.cpp -- TEST DOUBLE for librccl, used by tests/test_gpu_dist.py only (never by the product: libgnnagg.so loads the real
// librccl unless the test hook GNNAGG_RCCL_LIB names another library).
//
// RCCL needs one GPU per rank, the test box has one GPU.  This library implements the eight entry points dist_rccl.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString) for
// ranks that are PROCESSES SHARING ONE GPU: a message travels device -> a mailbox file in /dev/shm -> device.  What it keeps of
// the real thing is the contract the C-ABI step relies on: point-to-point messages matched per (source, destination) in posting
// order, byte counts that must agree on both ends, a group that completes all of its sends and receives, stream order (the
// group waits for the work enqueued before it and is complete when ncclGroupEnd returns -- stricter than RCCL, never weaker).
// It is synchronous, so it cannot be captured into a HIP graph.  With it the real step code -- pack kernel, per-stage grouped
// sends / receives to SEVERAL peers, offsets, events, the halo-source passes -- runs at world 2-4 on the one-GPU box.
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
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr size_t kHeader = 4096;
constexpr double kTimeoutS = 120.0;

struct BoxHeader {
    std::atomic<uint64_t> written;  // messages the sender has published
    std::atomic<uint64_t> read;     // messages the receiver has consumed
    uint64_t bytes;                 // size of message number `written`
};

struct Box {   // one direction of one pair: file /dev/shm/fakerccl_<token>_<src>_<dst>
    int fd = -1;
    char *map = nullptr;
    size_t mapped = 0;
    uint64_t count = 0;  // messages this end has sent / received
    BoxHeader *hdr() { return reinterpret_cast<BoxHeader *>(map); }
    bool remap(size_t need)
    {
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size_t sz = (size_t)st.st_size;
        if (sz < need) {
            if (ftruncate(fd, (off_t)need) != 0) return false;
            sz = need;
        }
        if (sz == mapped) return true;
        if (map) munmap(map, mapped);
        map = static_cast<char *>(mmap(nullptr, sz, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        mapped = sz;
        return map != MAP_FAILED;
    }
    void close_box()
    {
        if (map && map != MAP_FAILED) munmap(map, mapped);
        if (fd >= 0) close(fd);
        map = nullptr; fd = -1;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_ops;
}  // namespace

struct ncclComm {
    int rank = 0, world = 1;
    std::string token;
    std::vector<Box> out, in;   // out[p]: rank -> p, in[p]: p -> rank
    std::string path(int src, int dst) const { return "/dev/shm/fakerccl_" + token + "_" + std::to_string(src) + "_" + std::to_string(dst); }
};

static ncclResult_t do_send(ncclComm *c, const Op &o)
{
    Box &b = c->out[o.peer];
    const double t0 = now_s();
    while (b.hdr()->read.load(std::memory_order_acquire) != b.count) {   // the previous message to this peer is still unread
        if (now_s() - t0 > kTimeoutS) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (!b.remap(kHeader + o.bytes)) return ncclSystemError;
    if (o.bytes && hipMemcpy(b.map + kHeader, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    b.hdr()->bytes = o.bytes;
    b.hdr()->written.store(++b.count, std::memory_order_release);
    return ncclSuccess;
}

static ncclResult_t do_recv(ncclComm *c, const Op &o)
{
    Box &b = c->in[o.peer];
    const uint64_t want = b.count + 1;
    const double t0 = now_s();
    while (b.hdr()->written.load(std::memory_order_acquire) < want) {
        if (now_s() - t0 > kTimeoutS) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (b.hdr()->bytes != o.bytes) {
        fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, o.bytes, o.peer, (unsigned long long)b.hdr()->bytes);
        return ncclInvalidArgument;
    }
    if (!b.remap(kHeader + o.bytes)) return ncclSystemError;
    if (o.bytes && hipMemcpy(o.buf, b.map + kHeader, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    b.count = want;
    b.hdr()->read.store(want, std::memory_order_release);
    return ncclSuccess;
}

static ncclResult_t flush_ops()
{
    std::vector<std::pair<ncclComm *, Op>> ops;
    ops.swap(g_ops);
    for (auto &e : ops)   // stream order: everything enqueued before the group has run
        if (hipStreamSynchronize(e.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (auto &e : ops)
        if (e.second.send) { const ncclResult_t r = do_send(e.first, e.second); if (r != ncclSuccess) return r; }
    for (auto &e : ops)
        if (!e.second.send) { const ncclResult_t r = do_recv(e.first, e.second); if (r != ncclSuccess) return r; }
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
    ncclComm *c = new ncclComm;
    c->rank = rank; c->world = nranks;
    c->token.assign(id.internal, strnlen(id.internal, sizeof(id.internal)));
    c->out.resize(nranks); c->in.resize(nranks);
    for (int p = 0; p < nranks; ++p) {   // my outgoing boxes: created under a temporary name, published by rename
        const std::string fin = c->path(rank, p), tmp = fin + ".tmp";
        const int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)kHeader) != 0 || rename(tmp.c_str(), fin.c_str()) != 0) { delete c; return ncclSystemError; }
        c->out[p].fd = fd;
        if (!c->out[p].remap(kHeader)) { delete c; return ncclSystemError; }
    }
    const double t0 = now_s();
    for (int p = 0; p < nranks; ++p) {   // my incoming boxes: created by the peers
        int fd;
        while ((fd = open(c->path(p, rank).c_str(), O_RDWR)) < 0) {
            if (now_s() - t0 > kTimeoutS) { delete c; return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        c->in[p].fd = fd;
        if (!c->in[p].remap(kHeader)) { delete c; return ncclSystemError; }
    }
    *comm = c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    for (int p = 0; p < comm->world; ++p) {
        comm->out[p].close_box();
        comm->in[p].close_box();
        unlink(comm->path(comm->rank, p).c_str());
    }
    delete comm;
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
        case ncclSystemError: return "fake_rccl: mailbox error or a peer did not answer in time";
        case ncclInvalidArgument: return "fake_rccl: invalid argument or mismatched message size";
        case ncclUnhandledCudaError: return "fake_rccl: HIP error";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"

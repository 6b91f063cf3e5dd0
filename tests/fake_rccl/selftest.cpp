// selftest.cpp -- checks the librccl TEST DOUBLE itself (fake_rccl.cpp) before anything relies on it: started once per rank by
// tests/test_gpu_fake_rccl.py (RANK / WORLD_SIZE in the environment, --idfile for the unique id), all ranks on the one GPU.
//
//   rounds  R all-to-all-v rounds back to back with NO host synchronisation between them: per round a kernel writes a round-dependent
//           pattern into the send buffer, the receive buffer is poisoned, grouped ncclSend / ncclRecv to every peer, a kernel counts the
//           wrong words -- all on one stream.  Round r + 1's pattern kernel may only run when round r's sends have completed on the
//           stream, round r's checker only behind its receives: what the `ready` / `done` counters are for.
//   async   rank 1's groups are delayed by 300 ms on the DEVICE.  Rank 0's ncclGroupEnd must return long before that, and a copy of
//           its receive buffer taken on ANOTHER stream right after the return must still show the poison: the double does not wait for
//           the data on the host -- the property the synchronous round-1..5 double lacked.  After a stream synchronise the data is there.
//   graph   one round captured into a HIP graph (the round number lives on the device), replayed 5 times by every rank.
//   handles what hipIpcGetMemHandle returns for the same allocation twice, and for a new allocation at a recycled address (printed;
//           decides how the double caches exports).
//
// Prints one JSON line; exit code 0 only when every check passed.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" void fakeRcclSetDelayUs(ncclComm_t comm, double us);

#define HCK(x)                                                                                          \
    do {                                                                                                \
        hipError_t _e = (x);                                                                            \
        if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); exit(2); }      \
    } while (0)
#define NCK(x)                                                                                          \
    do {                                                                                                \
        ncclResult_t _r = (x);                                                                          \
        if (_r != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(_r)); exit(2); }    \
    } while (0)

__device__ __forceinline__ unsigned pattern(unsigned src, unsigned dst, unsigned round, unsigned i)
{
    return (src * 0x9e3779b1u) ^ (dst * 0x85ebca6bu) ^ (round * 0xc2b2ae35u) ^ (i * 2654435761u + 12345u);
}

// seg[p] .. seg[p + 1]: the words addressed to (send side) / coming from (receive side) rank p
__global__ void k_fill(unsigned *send, const long long *seg, int world, int rank, unsigned *round_ctr, int bump)
{
    const unsigned round = *round_ctr + (bump ? 1u : 0u);
    for (int p = 0; p < world; ++p)
        for (long long i = seg[p] + blockIdx.x * blockDim.x + threadIdx.x; i < seg[p + 1]; i += (long long)gridDim.x * blockDim.x)
            send[i] = pattern(rank, p, round, (unsigned)(i - seg[p]));
}
__global__ void k_bump(unsigned *round_ctr) { *round_ctr += 1; }
__global__ void k_check(const unsigned *recv, const long long *seg, int world, int rank, const unsigned *round_ctr, unsigned long long *bad)
{
    const unsigned round = *round_ctr;
    unsigned long long n = 0;
    for (int p = 0; p < world; ++p)
        for (long long i = seg[p] + blockIdx.x * blockDim.x + threadIdx.x; i < seg[p + 1]; i += (long long)gridDim.x * blockDim.x)
            n += recv[i] != pattern(p, rank, round, (unsigned)(i - seg[p]));
    if (n) atomicAdd(bad, n);
}

static long long words_between(int src, int dst, int world) { return src == dst ? 0 : 20000 + 7919LL * ((src * world + dst) % 13) + (src == 0 ? 300000 : 0); }

struct Bufs {
    int rank, world;
    std::vector<long long> sseg, rseg;
    long long *d_sseg, *d_rseg;
    unsigned *d_send, *d_recv, *d_round;
    unsigned long long *d_bad;
};

static void exchange(ncclComm_t comm, const Bufs &b, hipStream_t st)
{
    NCK(ncclGroupStart());
    for (int p = 0; p < b.world; ++p) {
        if (p == b.rank) continue;
        if (b.sseg[p + 1] > b.sseg[p]) NCK(ncclSend(b.d_send + b.sseg[p], (size_t)(b.sseg[p + 1] - b.sseg[p]) * 4, ncclInt8, p, comm, st));
        if (b.rseg[p + 1] > b.rseg[p]) NCK(ncclRecv(b.d_recv + b.rseg[p], (size_t)(b.rseg[p + 1] - b.rseg[p]) * 4, ncclInt8, p, comm, st));
    }
    NCK(ncclGroupEnd());
}

static void one_round(ncclComm_t comm, const Bufs &b, hipStream_t st)
{
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, b.d_send, b.d_sseg, b.world, b.rank, b.d_round, 1);
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, st, b.d_round);
    HCK(hipMemsetAsync(b.d_recv, 0xff, (size_t)b.rseg[b.world] * 4, st));
    exchange(comm, b, st);
    hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, st, b.d_recv, b.d_rseg, b.world, b.rank, b.d_round, b.d_bad);
}

static unsigned long long read_bad(const Bufs &b)
{
    unsigned long long v = 0;
    HCK(hipMemcpy(&v, b.d_bad, sizeof(v), hipMemcpyDeviceToHost));
    return v;
}

int main(int argc, char **argv)
{
    const int rank = atoi(getenv("RANK") ? getenv("RANK") : "0"), world = atoi(getenv("WORLD_SIZE") ? getenv("WORLD_SIZE") : "1");
    std::string idfile = "/tmp/fakerccl_selftest_id";
    int rounds = 20;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--idfile")) idfile = argv[i + 1];
        if (!strcmp(argv[i], "--rounds")) rounds = atoi(argv[i + 1]);
    }
    HCK(hipSetDevice(0));
    // ---- handles
    int same_twice = -1, same_after_realloc = -1, recycled = 0;
    {
        void *a = nullptr;
        HCK(hipMalloc(&a, 1 << 20));
        hipIpcMemHandle_t h1, h2, h3;
        HCK(hipIpcGetMemHandle(&h1, a));
        HCK(hipIpcGetMemHandle(&h2, a));
        same_twice = memcmp(&h1, &h2, sizeof(h1)) == 0;
        HCK(hipFree(a));
        void *b = nullptr;
        HCK(hipMalloc(&b, 1 << 20));
        recycled = a == b;
        HCK(hipIpcGetMemHandle(&h3, b));
        same_after_realloc = memcmp(&h1, &h3, sizeof(h1)) == 0;
        HCK(hipFree(b));
    }
    // ---- communicator: rank 0 draws the id and passes it through a file
    ncclUniqueId id;
    if (rank == 0) {
        NCK(ncclGetUniqueId(&id));
        FILE *f = fopen((idfile + ".tmp").c_str(), "wb");
        if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 2;
        fclose(f);
        if (rename((idfile + ".tmp").c_str(), idfile.c_str()) != 0) return 2;
    } else {
        for (int tries = 0;; ++tries) {
            FILE *f = fopen(idfile.c_str(), "rb");
            if (f) {
                const bool ok = fread(&id, sizeof(id), 1, f) == 1;
                fclose(f);
                if (ok) break;
            }
            if (tries > 60000) return 2;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    ncclComm_t comm;
    NCK(ncclCommInitRank(&comm, world, id, rank));
    Bufs b;
    b.rank = rank; b.world = world;
    b.sseg.assign(world + 1, 0); b.rseg.assign(world + 1, 0);
    for (int p = 0; p < world; ++p) {
        b.sseg[p + 1] = b.sseg[p] + words_between(rank, p, world);
        b.rseg[p + 1] = b.rseg[p] + words_between(p, rank, world);
    }
    HCK(hipMalloc((void **)&b.d_sseg, sizeof(long long) * (world + 1)));
    HCK(hipMalloc((void **)&b.d_rseg, sizeof(long long) * (world + 1)));
    HCK(hipMemcpy(b.d_sseg, b.sseg.data(), sizeof(long long) * (world + 1), hipMemcpyHostToDevice));
    HCK(hipMemcpy(b.d_rseg, b.rseg.data(), sizeof(long long) * (world + 1), hipMemcpyHostToDevice));
    HCK(hipMalloc((void **)&b.d_send, (size_t)std::max<long long>(b.sseg[world], 1) * 4));
    HCK(hipMalloc((void **)&b.d_recv, (size_t)std::max<long long>(b.rseg[world], 1) * 4));
    HCK(hipMalloc((void **)&b.d_round, 4));
    HCK(hipMalloc((void **)&b.d_bad, 8));
    HCK(hipMemset(b.d_round, 0, 4));
    HCK(hipMemset(b.d_bad, 0, 8));
    hipStream_t st, side;
    HCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HCK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    // ---- rounds: nothing but stream order between them
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < rounds; ++r) one_round(comm, b, st);
    const double enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HCK(hipStreamSynchronize(st));
    const unsigned long long bad_rounds = read_bad(b);
    // ---- async: rank 1 late by 300 ms on the device
    double group_end_ms = 0;
    unsigned long long poison_seen = 0, poison_total = 0, bad_async = 0;
    if (world > 1) {
        HCK(hipDeviceSynchronize());
        fakeRcclSetDelayUs(comm, rank == 1 ? 300000.0 : 0.0);
        hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, b.d_send, b.d_sseg, world, rank, b.d_round, 1);
        hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, st, b.d_round);
        HCK(hipMemsetAsync(b.d_recv, 0xff, (size_t)b.rseg[world] * 4, st));
        HCK(hipStreamSynchronize(st));    // the poison is in place; from here on only the exchange writes d_recv
        const auto g0 = std::chrono::steady_clock::now();
        exchange(comm, b, st);
        group_end_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();
        if (rank != 1) {   // what rank 1 sends cannot have arrived: its "in place" signal sits behind 300 ms of spinning
            const long long n = b.rseg[2] - b.rseg[1];
            std::vector<unsigned> peek((size_t)n);
            HCK(hipMemcpyAsync(peek.data(), b.d_recv + b.rseg[1], (size_t)n * 4, hipMemcpyDeviceToHost, side));
            HCK(hipStreamSynchronize(side));
            poison_total = (unsigned long long)n;
            for (unsigned v : peek) poison_seen += v == 0xffffffffu;
        }
        hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, st, b.d_recv, b.d_rseg, world, rank, b.d_round, b.d_bad);
        HCK(hipStreamSynchronize(st));
        bad_async = read_bad(b) - bad_rounds;
        fakeRcclSetDelayUs(comm, 0.0);
    }
    // ---- graph: capture one round, replay it 5 times
    unsigned long long bad_graph = 0;
    int replays = 0;
    {
        HCK(hipDeviceSynchronize());
        hipGraph_t g;
        hipGraphExec_t ge;
        HCK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        one_round(comm, b, st);
        HCK(hipStreamEndCapture(st, &g));
        HCK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (; replays < 5; ++replays) HCK(hipGraphLaunch(ge, st));
        HCK(hipStreamSynchronize(st));
        bad_graph = read_bad(b) - bad_rounds - bad_async;
        unsigned round_now = 0;
        HCK(hipMemcpy(&round_now, b.d_round, 4, hipMemcpyDeviceToHost));
        if ((int)round_now != rounds + (world > 1 ? 1 : 0) + replays) bad_graph += 1000000;   // every replay ran its own round
        HCK(hipGraphExecDestroy(ge));
        HCK(hipGraphDestroy(g));
    }
    NCK(ncclCommDestroy(comm));
    const bool async_ok = world == 1 || rank == 1 || (poison_seen == poison_total && group_end_ms < 150.0);
    const bool ok = bad_rounds == 0 && bad_async == 0 && bad_graph == 0 && async_ok;
    printf("{\"rank\": %d, \"world\": %d, \"rounds\": %d, \"bad_words_rounds\": %llu, \"enqueue_ms\": %.2f, \"group_end_ms_with_late_peer\": %.2f, "
           "\"poison_words_seen_after_group_end\": %llu, \"poison_words_expected\": %llu, \"bad_words_async\": %llu, \"graph_replays\": %d, "
           "\"bad_words_graph\": %llu, \"ipc_handle_same_twice\": %d, \"ipc_handle_same_after_realloc\": %d, \"address_recycled\": %d, \"ok\": %s}\n",
           rank, world, rounds, bad_rounds, enqueue_ms, group_end_ms, poison_seen, poison_total, bad_async, replays, bad_graph, same_twice,
           same_after_realloc, recycled, ok ? "true" : "false");
    return ok ? 0 : 1;
}

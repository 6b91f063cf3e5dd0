// dist_rccl.cpp -- RCCL transport of the row-partitioned (multi-GPU) path behind the C-ABI (gnnagg.h Section D).
//
// The reference has no multi-GPU aggregation (Figure9/main.cu:19 asserts GPUNUM == 1; its NCCL calls are commented out,
// include/util.h:25,42,72); SURVEY.md 8(e) is the specification: one-time all-to-all of request lists, then per
// aggregation a pack kernel + ncclGroupStart / 7 x (ncclSend, ncclRecv) / ncclGroupEnd on the caller's stream -- over the
// point-to-point xGMI mesh every pairwise message rides its own link, so a grouped send/recv IS the all-to-all-v.
//
// librccl is loaded lazily with dlopen: libgnnagg.so has no link-time dependency on it (the single-GPU path never needs
// it), and a process that already holds a copy -- torch ships one -- keeps using that copy (RTLD_NOLOAD first).
#include <dlfcn.h>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace gnnagg {

struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string path;        // file the eight entry points really come from (dladdr), for gnnagg_dist_transport_info
    bool overridden = false; // loaded through GNNAGG_RCCL_LIB (a test double), not the system's librccl
};

static RcclApi *rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // test hook: another library with the same eight entry points (tests/fake_rccl: ranks as processes sharing one GPU, so that
        // the step's multi-peer code runs on a one-GPU box).  RTLD_LOCAL: its symbols must not shadow the real librccl torch maps
        if (const char *over = getenv("GNNAGG_RCCL_LIB")) {
            if (*over) api.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            api.overridden = api.h != nullptr;
        }
        if (!api.h)
        for (const char *n : names)
            if ((api.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;  // a copy this process already mapped
        if (!api.h)
            for (const char *n : names)
                if ((api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.h) return;
#define LOAD(sym) api.sym = reinterpret_cast<decltype(api.sym)>(dlsym(api.h, "nccl" #sym))
        LOAD(GetUniqueId); LOAD(CommInitRank); LOAD(CommDestroy); LOAD(GroupStart); LOAD(GroupEnd); LOAD(Send); LOAD(Recv);
        LOAD(GetErrorString);
#undef LOAD
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv)
            api.h = nullptr;
        Dl_info info;
        if (api.h && dladdr(reinterpret_cast<void *>(api.Send), &info) && info.dli_fname) api.path = info.dli_fname;
    });
    return api.h ? &api : nullptr;
}

#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess)                                                                                 \
            return fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(_r) : "rccl error")); \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int device = -1;   // HIP device that was current when the communicator was made (ncclCommInitRank binds to it)
};

static std::mutex g_comm_mu;
static std::set<Comm *> g_comms;

static Comm *lookup_comm(gnnagg_comm c)
{
    std::lock_guard<std::mutex> lk(g_comm_mu);
    Comm *p = reinterpret_cast<Comm *>(c);
    return g_comms.count(p) ? p : nullptr;
}

}  // namespace gnnagg

using namespace gnnagg;

#pragma GCC visibility push(default)
extern "C" {

int gnnagg_dist_unique_id(char *id128)
{
    if (!id128) return fail(GNNAGG_ERR_ARG, "null id buffer");
    RcclApi *R = rccl();
    if (!R) return fail(GNNAGG_ERR_STATE, "librccl could not be loaded");
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == GNNAGG_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return GNNAGG_OK;
}

int gnnagg_dist_comm_create(const char *id128, int rank, int world, gnnagg_comm *out)
{
    if (!out) return fail(GNNAGG_ERR_ARG, "null output communicator");
    *out = 0;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(GNNAGG_ERR_ARG, "bad communicator arguments");
    RcclApi *R = rccl();
    if (!R) return fail(GNNAGG_ERR_STATE, "librccl could not be loaded");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm;
    c->rank = rank; c->world = world;
    (void)hipGetDevice(&c->device);
    const ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);  // binds to the calling thread's current HIP device
    if (r != ncclSuccess) {
        delete c;
        return fail(GNNAGG_ERR_HIP, std::string("ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(r) : "rccl error"));
    }
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        g_comms.insert(c);
    }
    *out = reinterpret_cast<gnnagg_comm>(c);
    return GNNAGG_OK;
}

// Rendezvous through the file system, safe against what an earlier (crashed) launch left at the same path: ranks > 0 each
// publish a fresh random token (<path>.req.<rank>); rank 0 first removes whatever is at <path>, draws the id, collects the
// tokens of THIS launch and publishes {magic, world, id, tokens} atomically; a rank accepts an id file only when it carries the
// token it drew itself -- a stale file cannot, so nobody ever calls ncclCommInitRank with mismatched ids (which hangs forever).
// A stale request file costs a round: rank 0 answers it, its owner rejects the answer and republishes its token, rank 0 notices
// the changed token and publishes again.  Every wait is bounded by timeout_s.
static bool read_file(const std::string &p, void *buf, size_t n)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    const size_t got = fread(buf, 1, n, f);
    const bool exact = got == n && fgetc(f) == EOF;
    fclose(f);
    return exact;
}

static bool publish_file(const std::string &p, const void *buf, size_t n)
{
    const std::string tmp = p + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(buf, 1, n, f) == n;
    fclose(f);
    if (!ok || rename(tmp.c_str(), p.c_str()) != 0) { (void)unlink(tmp.c_str()); return false; }  // atomic: all bytes or no file
    return true;
}

static unsigned long long fresh_token()
{
    unsigned long long t = 0;
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&t, sizeof(t), 1, f) != 1) t = 0; fclose(f); }
    if (t == 0)
        t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ULL ^ ((unsigned long long)getpid() << 32);
    return t ? t : 1;
}

int gnnagg_dist_comm_create_from_file(const char *path, int rank, int world, int timeout_s, gnnagg_comm *out)
{
    if (!path || !out) return fail(GNNAGG_ERR_ARG, "bad communicator arguments");
    *out = 0;
    if (world < 1 || rank < 0 || rank >= world) return fail(GNNAGG_ERR_ARG, "bad communicator arguments");
    static constexpr unsigned long long kMagic = 0x31444947414e4e47ULL;  // "GNNAGID1"
    const std::string p(path);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s > 0 ? timeout_s : 120);
    const size_t n_words = 2 + GNNAGG_UNIQUE_ID_BYTES / 8 + (size_t)(world - 1);  // magic, world, id, one token per rank > 0
    std::vector<unsigned long long> rec(n_words, 0);
    char *id = reinterpret_cast<char *>(&rec[2]);
    auto req_path = [&](int r) { return p + ".req." + std::to_string(r); };
    if (rank == 0) {
        (void)unlink(p.c_str());  // whatever an earlier launch left here is not ours
        int rc = gnnagg_dist_unique_id(id);
        if (rc) return rc;
        rec[0] = kMagic;
        rec[1] = (unsigned long long)world;
        std::vector<unsigned long long> seen((size_t)world, 0);
        bool published = false;
        for (;;) {
            bool all = true, changed = false;
            for (int r = 1; r < world; ++r) {
                unsigned long long t = 0;
                if (!read_file(req_path(r), &t, sizeof(t)) || t == 0) { all = false; continue; }
                if (t != seen[r]) { seen[r] = t; changed = true; }
            }
            if (all && (changed || !published)) {
                for (int r = 1; r < world; ++r) rec[2 + GNNAGG_UNIQUE_ID_BYTES / 8 + (size_t)(r - 1)] = seen[r];
                if (!publish_file(p, rec.data(), n_words * 8)) return fail(GNNAGG_ERR_IO, "cannot publish " + p);
                published = true;
            }
            // done when every request file is gone again: its owner accepted the id file and removed it
            bool gone = published;
            for (int r = 1; r < world && gone; ++r) gone = access(req_path(r).c_str(), F_OK) != 0;
            if (gone) break;
            if (std::chrono::steady_clock::now() > t_end) return fail(GNNAGG_ERR_IO, "timed out waiting for the other ranks at " + p);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    } else {
        const unsigned long long mine = fresh_token();
        const std::string rq = req_path(rank);
        if (!publish_file(rq, &mine, sizeof(mine))) return fail(GNNAGG_ERR_IO, "cannot write " + rq);
        for (;;) {
            if (read_file(p, rec.data(), n_words * 8) && rec[0] == kMagic && rec[1] == (unsigned long long)world &&
                rec[2 + GNNAGG_UNIQUE_ID_BYTES / 8 + (size_t)(rank - 1)] == mine)
                break;
            if (std::chrono::steady_clock::now() > t_end) { (void)unlink(rq.c_str()); return fail(GNNAGG_ERR_IO, "timed out waiting for " + p); }
            unsigned long long t = 0;   // (somebody cleaning the directory must not strand this rank: keep the request in place)
            if (!read_file(rq, &t, sizeof(t)) || t != mine) (void)publish_file(rq, &mine, sizeof(mine));
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        (void)unlink(rq.c_str());
    }
    return gnnagg_dist_comm_create(id, rank, world, out);
}

int gnnagg_dist_comm_destroy(gnnagg_comm h)
{
    Comm *c;
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        c = reinterpret_cast<Comm *>(h);
        if (!g_comms.count(c)) return fail(GNNAGG_ERR_ARG, "invalid or destroyed communicator");
        g_comms.erase(c);
    }
    RcclApi *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    return GNNAGG_OK;
}

int gnnagg_dist_comm_info(gnnagg_comm h, int *rank, int *world)
{
    Comm *c = lookup_comm(h);
    if (!c) return fail(GNNAGG_ERR_ARG, "invalid or destroyed communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return GNNAGG_OK;
}

int gnnagg_dist_transport_info(gnnagg_comm h, char *library_path, int library_path_cap, char *pci_bus_id, int pci_bus_id_cap, int *is_override)
{
    Comm *c = lookup_comm(h);
    if (!c) return fail(GNNAGG_ERR_ARG, "invalid or destroyed communicator");
    RcclApi *R = rccl();
    if (library_path && library_path_cap > 0) snprintf(library_path, (size_t)library_path_cap, "%s", R ? R->path.c_str() : "");
    if (is_override) *is_override = R && R->overridden ? 1 : 0;
    if (pci_bus_id && pci_bus_id_cap > 0) {
        pci_bus_id[0] = 0;
        if (c->device >= 0 && hipDeviceGetPCIBusId(pci_bus_id, pci_bus_id_cap, c->device) != hipSuccess) pci_bus_id[0] = 0;
    }
    return GNNAGG_OK;
}

int gnnagg_dist_alltoallv(gnnagg_comm h, const void *d_send, const long long *h_send_counts, void *d_recv,
                          const long long *h_recv_counts, int elem_bytes, void *hip_stream)
{
    Comm *c = lookup_comm(h);
    if (!c) return fail(GNNAGG_ERR_ARG, "invalid or destroyed communicator");
    if (!h_send_counts || !h_recv_counts || elem_bytes <= 0) return fail(GNNAGG_ERR_ARG, "bad alltoallv arguments");
    RcclApi *R = rccl();
    if (!R) return fail(GNNAGG_ERR_STATE, "librccl could not be loaded");
    hipStream_t stream = (hipStream_t)hip_stream;
    const char *sp = static_cast<const char *>(d_send);
    char *rp = static_cast<char *>(d_recv);
    size_t soff = 0, roff = 0;
    // the part a rank keeps for itself is a device-to-device copy on the same stream (no self send/recv in the group)
    std::vector<size_t> so((size_t)c->world), ro((size_t)c->world);
    for (int p = 0; p < c->world; ++p) {
        if (h_send_counts[p] < 0 || h_recv_counts[p] < 0) return fail(GNNAGG_ERR_ARG, "negative alltoallv count");
        so[p] = soff; ro[p] = roff;
        soff += (size_t)h_send_counts[p] * elem_bytes;
        roff += (size_t)h_recv_counts[p] * elem_bytes;
    }
    if ((soff > 0 && !d_send) || (roff > 0 && !d_recv)) return fail(GNNAGG_ERR_ARG, "null alltoallv buffer");
    if (h_send_counts[c->rank] != h_recv_counts[c->rank]) return fail(GNNAGG_ERR_ARG, "alltoallv: self send and receive counts differ");
    if (h_send_counts[c->rank] > 0) {
        const hipError_t e = hipMemcpyAsync(rp + ro[c->rank], sp + so[c->rank], (size_t)h_send_counts[c->rank] * elem_bytes,
                                            hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return fail(GNNAGG_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    }
    if (c->world == 1) return GNNAGG_OK;
    RCCL_TRY(R->GroupStart());
    // an error inside the bracket must still close it: a thread left with an open group queues every later RCCL call
    // (torch's too: it shares this librccl) and never issues it
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
        if (p == c->rank) continue;
        if (h_send_counts[p] > 0) r = R->Send(sp + so[p], (size_t)h_send_counts[p] * elem_bytes, ncclInt8, p, c->comm, stream);
        if (r == ncclSuccess && h_recv_counts[p] > 0) r = R->Recv(rp + ro[p], (size_t)h_recv_counts[p] * elem_bytes, ncclInt8, p, c->comm, stream);
    }
    if (r != ncclSuccess) {
        (void)R->GroupEnd();
        return fail(GNNAGG_ERR_HIP, std::string("ncclSend / ncclRecv: ") + (R->GetErrorString ? R->GetErrorString(r) : "rccl error"));
    }
    RCCL_TRY(R->GroupEnd());
    return GNNAGG_OK;
}

int gnnagg_dist_halo_exchange(gnnagg_comm h, const float *d_x_local, const int *d_send_ids, const long long *h_send_rows,
                              const long long *h_recv_rows, int feat, float *d_send_buf, float *d_x_halo, void *hip_stream)
{
    Comm *c = lookup_comm(h);
    if (!c) return fail(GNNAGG_ERR_ARG, "invalid or destroyed communicator");
    if (!h_send_rows || !h_recv_rows || feat <= 0) return fail(GNNAGG_ERR_ARG, "bad halo_exchange arguments");
    long long n_send = 0;
    for (int p = 0; p < c->world; ++p) n_send += h_send_rows[p];
    if (n_send > 0x7fffffffLL) return fail(GNNAGG_ERR_ARG, "too many halo rows to pack");
    if (n_send > 0) {
        const int rc = launch_pack_rows(d_x_local, d_send_ids, (int)n_send, feat, d_send_buf, hip_stream);
        if (rc) return rc;
    }
    return gnnagg_dist_alltoallv(h, d_send_buf, h_send_rows, d_x_halo, h_recv_rows, feat * (int)sizeof(float), hip_stream);
}

int gnnagg_pack_rows2(const float *d_x, const float *d_att, const int *d_ids, int n, int feat, int att_width, float *d_out, void *hip_stream)
{
    if (n < 0 || feat <= 0 || att_width <= 0 || (n > 0 && (!d_x || !d_att || !d_ids || !d_out))) return fail(GNNAGG_ERR_ARG, "bad pack_rows2 arguments");
    return launch_pack_rows2(d_x, d_att, d_ids, n, feat, att_width, d_out, hip_stream);
}

int gnnagg_unpack_rows2(const float *d_in, int n, int feat, int att_width, float *d_x_out, float *d_att_out, void *hip_stream)
{
    if (n < 0 || feat <= 0 || att_width <= 0 || (n > 0 && (!d_in || !d_x_out || !d_att_out))) return fail(GNNAGG_ERR_ARG, "bad unpack_rows2 arguments");
    return launch_unpack_rows2(d_in, n, feat, att_width, d_x_out, d_att_out, hip_stream);
}

// ---------------------------------------------------------------------------------------------- one call per step
// The row-partitioned aggregation step behind ONE host call: on the caller's stream the local-source pass; on
// the step's communication stream, forked and joined with events, the pack kernel and the grouped send / recv; then the
// halo-source pass, which adds to what the local pass wrote.  Everything is enqueued asynchronously -- stream operations only,
// so a warm step can be captured into a HIP graph like any fork / join of two streams -- and nothing is allocated per step.
// A rank without peers (world 1) or without halo rows never creates the second stream: its step is the local pass alone.
//
// Staged form (round 4): the halo arrives in S stages, every stage one grouped send / recv with an event behind it, and the
// halo-source edges are split by the stage their source row arrives in: the pass over stage s's edges runs while stage s + 1 is
// on the links.  Buffers are stage-major, so a stage's rows are contiguous on both sides and every stage is a plain
// all-to-all-v on a sub-range.  On the point-to-point xGMI mesh every peer pair has its own link: a stage that talks to ONE
// peer uses one link of seven, so the default plan (dist.py, "stripe") gives every stage a slice of EVERY peer's rows.
namespace gnnagg {
struct DistStep {
    gnnagg_comm comm = 0;
    int world = 1, rank = 0, n_stages = 1;
    gnnagg_handle agg_local = 0;
    std::vector<gnnagg_handle> agg_remote;                 // [n_stages], 0 = no halo-source edges in that stage
    const int *d_send_ids = nullptr;
    std::vector<long long> send_rows, recv_rows;           // [n_stages * world]
    std::vector<long long> stage_send0, stage_recv0;       // [n_stages + 1] first row of every stage in the send buffer / halo tail
    long long n_send = 0, n_recv = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_stage;                      // [n_stages]: stage s has landed (the last one is the join)
    // (world 1 with rows addressed to itself -- the self part of an all-to-all-v is a stream-ordered copy -- still takes the staged path:
    // tests/test_gpu_dist.py drives every stage of it on one GPU that way)
    bool exchanging() const { return comm != 0 && (n_send > 0 || n_recv > 0); }
    bool any_remote() const { for (gnnagg_handle h : agg_remote) if (h) return true; return false; }
};
static std::mutex g_step_mu;
static std::set<DistStep *> g_steps;
static DistStep *lookup_step(gnnagg_dist_step_t h)
{
    std::lock_guard<std::mutex> lk(g_step_mu);
    DistStep *p = reinterpret_cast<DistStep *>(h);
    return g_steps.count(p) ? p : nullptr;
}
#define HIPD_TRY(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t _e = (expr);                                                                                     \
        if (_e != hipSuccess) return fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)
static int step_fork(DistStep *st, hipStream_t stream)
{
    if (!st->comm_stream) {  // first exchanging step: never inside a capture of a warm step
        HIPD_TRY(hipStreamCreateWithFlags(&st->comm_stream, hipStreamNonBlocking));
        HIPD_TRY(hipEventCreateWithFlags(&st->ev_fork, hipEventDisableTiming));
        st->ev_stage.assign((size_t)st->n_stages, nullptr);
        for (int s = 0; s < st->n_stages; ++s) HIPD_TRY(hipEventCreateWithFlags(&st->ev_stage[(size_t)s], hipEventDisableTiming));
    }
    HIPD_TRY(hipEventRecord(st->ev_fork, stream));            // x_local is ready where the caller's stream stands now
    HIPD_TRY(hipStreamWaitEvent(st->comm_stream, st->ev_fork, 0));
    return GNNAGG_OK;
}
// An error between fork and join: the caller's stream still joins the communication stream (whatever was enqueued there before the
// failure), so a stream capture is never left with an unjoined fork and the buffers are not reused under an exchange in flight.
// The error message of `rc` is kept.
static int step_abandon(DistStep *st, hipStream_t stream, int rc)
{
    const std::string msg = gnnagg_last_error();
    hipEvent_t ev = st->ev_stage.empty() ? nullptr : st->ev_stage.back();
    if (ev && hipEventRecord(ev, st->comm_stream) == hipSuccess) (void)hipStreamWaitEvent(stream, ev, 0);
    return fail(rc, msg);
}
// A HIP call between fork and join that fails takes the abandon path too (the fork is ALWAYS joined, also during a capture) ...
#define HIPD_STEP(expr)                                                                                                                   \
    do {                                                                                                                                  \
        hipError_t _e = (expr);                                                                                                           \
        if (_e != hipSuccess) return step_abandon(st, stream, fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)));   \
    } while (0)
// ... and a failed join of one stage is recorded while the remaining stages are still joined
#define HIPD_JOIN(expr)                                                                                                   \
    do {                                                                                                                  \
        hipError_t _e = (expr);                                                                                           \
        if (_e != hipSuccess && !rc) rc = fail(GNNAGG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)
// stage s of the exchange on the communication stream, rows of `row_floats` floats, then its event
static int step_exchange_stage(DistStep *st, int s, const float *d_send_buf, float *d_recv, int row_floats)
{
    const size_t w = (size_t)st->world;
    const int rc = gnnagg_dist_alltoallv(st->comm, d_send_buf + (size_t)st->stage_send0[(size_t)s] * row_floats, &st->send_rows[(size_t)s * w],
                                         d_recv + (size_t)st->stage_recv0[(size_t)s] * row_floats, &st->recv_rows[(size_t)s * w],
                                         row_floats * (int)sizeof(float), st->comm_stream);
    return rc;
}
}  // namespace gnnagg

int gnnagg_dist_step_create_staged(gnnagg_comm comm, gnnagg_handle agg_local, int n_stages, const gnnagg_handle *agg_remote,
                                   const int *d_send_ids, const long long *h_send_rows, const long long *h_recv_rows,
                                   gnnagg_dist_step_t *out)
{
    if (!out) return fail(GNNAGG_ERR_ARG, "null output step");
    *out = 0;
    if (!agg_local) return fail(GNNAGG_ERR_ARG, "dist_step: no local aggregator");
    if (n_stages < 1 || n_stages > 64) return fail(GNNAGG_ERR_ARG, "dist_step: 1 .. 64 stages");
    DistStep *st = new DistStep;
    st->comm = comm; st->agg_local = agg_local; st->d_send_ids = d_send_ids; st->n_stages = n_stages;
    st->agg_remote.assign((size_t)n_stages, 0);
    if (agg_remote) st->agg_remote.assign(agg_remote, agg_remote + n_stages);
    st->stage_send0.assign((size_t)n_stages + 1, 0);
    st->stage_recv0.assign((size_t)n_stages + 1, 0);
    if (comm) {
        int rc = gnnagg_dist_comm_info(comm, &st->rank, &st->world);
        if (rc) { delete st; return rc; }
        if (!h_send_rows || !h_recv_rows) { delete st; return fail(GNNAGG_ERR_ARG, "dist_step: null row counts"); }
        const size_t n = (size_t)n_stages * st->world;
        st->send_rows.assign(h_send_rows, h_send_rows + n);
        st->recv_rows.assign(h_recv_rows, h_recv_rows + n);
        for (int s = 0; s < n_stages; ++s) {
            for (int p = 0; p < st->world; ++p) {
                const long long a = st->send_rows[(size_t)s * st->world + p], b = st->recv_rows[(size_t)s * st->world + p];
                if (a < 0 || b < 0) { delete st; return fail(GNNAGG_ERR_ARG, "dist_step: negative row count"); }
                st->n_send += a;
                st->n_recv += b;
            }
            st->stage_send0[(size_t)s + 1] = st->n_send;
            st->stage_recv0[(size_t)s + 1] = st->n_recv;
        }
        if (st->n_send > 0x7fffffffLL || st->n_recv > 0x7fffffffLL) { delete st; return fail(GNNAGG_ERR_ARG, "dist_step: too many halo rows"); }
        if (st->n_send > 0 && !d_send_ids) { delete st; return fail(GNNAGG_ERR_ARG, "dist_step: null send ids"); }
    }
    {
        std::lock_guard<std::mutex> lk(g_step_mu);
        g_steps.insert(st);
    }
    *out = reinterpret_cast<gnnagg_dist_step_t>(st);
    return GNNAGG_OK;
}

int gnnagg_dist_step_create(gnnagg_comm comm, gnnagg_handle agg_local, gnnagg_handle agg_remote, const int *d_send_ids,
                            const long long *h_send_rows, const long long *h_recv_rows, gnnagg_dist_step_t *out)
{
    return gnnagg_dist_step_create_staged(comm, agg_local, 1, &agg_remote, d_send_ids, h_send_rows, h_recv_rows, out);
}

int gnnagg_dist_step_info(gnnagg_dist_step_t h, int *n_stages, int *world)
{
    DistStep *st = lookup_step(h);
    if (!st) return fail(GNNAGG_ERR_ARG, "invalid or destroyed step");
    if (n_stages) *n_stages = st->n_stages;
    if (world) *world = st->world;
    return GNNAGG_OK;
}

int gnnagg_dist_step_destroy(gnnagg_dist_step_t h)
{
    DistStep *st;
    {
        std::lock_guard<std::mutex> lk(g_step_mu);
        st = reinterpret_cast<DistStep *>(h);
        if (!g_steps.count(st)) return fail(GNNAGG_ERR_ARG, "invalid or destroyed step");
        g_steps.erase(st);
    }
    if (st->comm_stream) {
        (void)hipStreamSynchronize(st->comm_stream);
        (void)hipStreamDestroy(st->comm_stream);
        (void)hipEventDestroy(st->ev_fork);
        for (hipEvent_t e : st->ev_stage) if (e) (void)hipEventDestroy(e);
    }
    delete st;
    return GNNAGG_OK;
}

int gnnagg_dist_step_gcn(gnnagg_dist_step_t h, const float *d_x_local, float *d_x_halo, float *d_send_buf, float *d_y, int feat, int reduce,
                         void *hip_stream)
{
    DistStep *st = lookup_step(h);
    if (!st) return fail(GNNAGG_ERR_ARG, "invalid or destroyed step");
    if (!d_x_local || !d_y || feat <= 0) return fail(GNNAGG_ERR_ARG, "bad dist_step arguments");
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc;
    const bool ex = st->exchanging();
    const int S = st->n_stages;
    if (ex) {
        if ((st->n_send > 0 && !d_send_buf) || (st->n_recv > 0 && !d_x_halo)) return fail(GNNAGG_ERR_ARG, "dist_step: null exchange buffer");
        if ((rc = step_fork(st, stream))) return rc;
        // ONE pack kernel for all stages (the send buffer is stage-major), then a grouped send / recv and an event per stage
        if (st->n_send > 0 && (rc = launch_pack_rows(d_x_local, st->d_send_ids, (int)st->n_send, feat, d_send_buf, st->comm_stream)))
            return step_abandon(st, stream, rc);
        for (int s = 0; s < S; ++s) {
            if ((rc = step_exchange_stage(st, s, d_send_buf, d_x_halo, feat))) return step_abandon(st, stream, rc);
            HIPD_STEP(hipEventRecord(st->ev_stage[(size_t)s], st->comm_stream));
        }
    }
    rc = gnnagg_set_stream(st->agg_local, stream);
    if (!rc) rc = gnnagg_gcn_run_ex(st->agg_local, d_x_local, d_y, feat, GNNAGG_MODE_BALANCED, reduce, 0);   // overlaps the exchange
    for (int s = 0; s < S; ++s) {
        // the caller's stream joins every stage (the last one is the join of the fork), on the error path too
        if (ex) HIPD_JOIN(hipStreamWaitEvent(stream, st->ev_stage[(size_t)s], 0));
        if (rc || !st->agg_remote[(size_t)s] || st->stage_recv0[(size_t)s + 1] == st->stage_recv0[(size_t)s]) continue;
        if (!(rc = gnnagg_set_stream(st->agg_remote[(size_t)s], stream)))
            rc = gnnagg_gcn_run_ex(st->agg_remote[(size_t)s], d_x_halo, d_y, feat, GNNAGG_MODE_BALANCED, reduce, GNNAGG_FLAG_ACCUMULATE);
    }
    return rc;
}

int gnnagg_dist_step_gat(gnnagg_dist_step_t h, float *d_x_ext, float *d_att_ext, int n_local, float *d_send_buf, float *d_recv_buf,
                         float *d_den, float *d_y, int feat, int heads, float slope, void *hip_stream)
{
    DistStep *st = lookup_step(h);
    if (!st) return fail(GNNAGG_ERR_ARG, "invalid or destroyed step");
    if (!d_x_ext || !d_att_ext || !d_y || !d_den || feat <= 0 || heads <= 0 || n_local < 0) return fail(GNNAGG_ERR_ARG, "bad dist_step arguments");
    hipStream_t stream = (hipStream_t)hip_stream;
    const int aw = 2 * heads, w = feat + aw;
    int rc;
    const bool ex = st->exchanging();
    const int S = st->n_stages;
    if (ex) {
        if ((st->n_send > 0 && !d_send_buf) || (st->n_recv > 0 && !d_recv_buf)) return fail(GNNAGG_ERR_ARG, "dist_step: null exchange buffer");
        if ((rc = step_fork(st, stream))) return rc;
        // ONE exchange per stage carries [feature row | attention terms] of every requested row; the received rows are split into
        // the halo tails of x_ext / att_ext stage by stage (same row order on both sides)
        if (st->n_send > 0 && (rc = launch_pack_rows2(d_x_ext, d_att_ext, st->d_send_ids, (int)st->n_send, feat, aw, d_send_buf, st->comm_stream)))
            return step_abandon(st, stream, rc);
        for (int s = 0; s < S; ++s) {
            if ((rc = step_exchange_stage(st, s, d_send_buf, d_recv_buf, w))) return step_abandon(st, stream, rc);
            const long long r0 = st->stage_recv0[(size_t)s], nr = st->stage_recv0[(size_t)s + 1] - r0;
            if (nr > 0 && (rc = launch_unpack_rows2(d_recv_buf + (size_t)r0 * w, (int)nr, feat, aw, d_x_ext + ((size_t)n_local + r0) * feat,
                                                    d_att_ext + ((size_t)n_local + r0) * aw, st->comm_stream)))
                return step_abandon(st, stream, rc);
            HIPD_STEP(hipEventRecord(st->ev_stage[(size_t)s], st->comm_stream));
        }
    }
    // numerators and denominators of the local-source edges while the exchange is in flight; every halo-source pass adds its own,
    // the last one divides (it runs for every row: a row without halo sources is divided all the same)
    if ((rc = gnnagg_set_stream(st->agg_local, stream))) return ex ? step_abandon(st, stream, rc) : rc;
    if (!st->any_remote()) {
        // (a step made without halo-source aggregators: one pass over X_ext.  It reads the halo tail, so it runs BEHIND the exchange)
        for (int s = 0; ex && s < S; ++s) HIPD_JOIN(hipStreamWaitEvent(stream, st->ev_stage[(size_t)s], 0));
        if (rc) return rc;
        return gnnagg_gat_run(st->agg_local, d_x_ext, d_att_ext, d_y, feat, heads, slope, GNNAGG_MODE_BALANCED, nullptr);
    }
    int last = -1;
    for (int s = 0; s < S; ++s) if (st->agg_remote[(size_t)s]) last = s;
    rc = gnnagg_gat_run_part(st->agg_local, d_x_ext, d_att_ext, d_y, feat, heads, slope, 1, d_den);
    for (int s = 0; s < S; ++s) {
        if (ex) HIPD_JOIN(hipStreamWaitEvent(stream, st->ev_stage[(size_t)s], 0));   // joined on the error path too
        if (rc || !st->agg_remote[(size_t)s]) continue;
        if (!(rc = gnnagg_set_stream(st->agg_remote[(size_t)s], stream)))
            rc = gnnagg_gat_run_part(st->agg_remote[(size_t)s], d_x_ext, d_att_ext, d_y, feat, heads, slope, s == last ? 2 : 3, d_den);
    }
    return rc;
}

}  // extern "C"
#pragma GCC visibility pop

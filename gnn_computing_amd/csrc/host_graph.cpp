// host_graph.cpp -- host-side graph preparation of libgnnagg (Section C/D of include/gnnagg.h).
//
// Counterpart of the reference's L1 layer (src/data.cu:4-139, include/graph_schedule.h:17-243),
// re-designed rather than transcribed: schedules are built count -> prefix-sum -> fill (so the
// output sizes are known before anything is written and rows can be processed in parallel),
// the text graph is parsed from one buffered read by a hand-rolled integer scanner instead of
// one fscanf per token, and the permutation is applied with a parallel scatter.
// Results are bit-identical to the reference routines (tests/test_host_graph.py pins them
// against the oracle and the golden vectors).
#include "common.h"

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace gnnagg {

// ------------------------------------------------------------------------------------ reorder
// new row i = old row map[i]; neighbor ids relabelled through reverse_map; in-row order kept
// (reference src/data.cu:4-29).
void reorder_csr(const int *ptr, const int *idx, const int *map, const int *rmap, int V, int *newptr, int *newidx)
{
    newptr[0] = 0;
    for (int i = 0; i < V; ++i) newptr[i + 1] = newptr[i] + (ptr[map[i] + 1] - ptr[map[i]]);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < V; ++i) {
        const int *src = idx + ptr[map[i]];
        int *dst = newidx + newptr[i];
        const int n = newptr[i + 1] - newptr[i];
        for (int j = 0; j < n; ++j) dst[j] = rmap[src[j]];
    }
}

// ---------------------------------------------------------------------------- neighbor grouping
// Row of degree d contributes ceil(d/NG) groups, the last one ragged (reference
// include/graph_schedule.h:91-126).  Returns G; fills ptr_out[G+1], target_out[G] when non-NULL.
int neighbor_grouping(const int *ptr, int ng, int V, int *ptr_out, int *target_out)
{
    std::vector<int> first(V + 1);
    first[0] = 0;
    for (int i = 0; i < V; ++i) {
        const int d = ptr[i + 1] - ptr[i];
        first[i + 1] = first[i] + (d + ng - 1) / ng;
    }
    const int G = first[V];
    if (ptr_out) {
        ptr_out[0] = 0;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < V; ++i) {
            const int n = first[i + 1] - first[i];
            for (int k = 0; k < n; ++k) {
                const int g = first[i] + k;
                ptr_out[g + 1] = std::min(ptr[i] + (k + 1) * ng, ptr[i + 1]);
                if (target_out) target_out[g] = i;
            }
        }
    } else if (target_out) {
        for (int i = 0; i < V; ++i)
            for (int g = first[i]; g < first[i + 1]; ++g) target_out[g] = i;
    }
    return G;
}

// ------------------------------------------------------------------------- locality (+ grouping)
// For each of par_num equal column ranges (last one extended to total_v), for each row, the
// sub-row of edges whose column falls in the range, optionally cut every ng edges (reference
// include/graph_schedule.h:17-63 and :156-211).  Pass 1 counts sub-row sizes per (partition,row),
// pass 2 scatters; both passes are row-parallel.
int locality_schedule(const int *ptr, const int *idx, const float *val, int par_num, int ng, int V, int total_v,
                      int *ptr_out, int *idx_out, float *val_out, int *target_out, int *eid_out)
{
    const int width = total_v / par_num;
    auto part_of = [&](int col) {
        // ranges [p*width, (p+1)*width), last one up to total_v; columns outside every range
        // (only possible when width == 0 or col >= total_v) are dropped like the reference does.
        if (col < 0 || col >= total_v) return -1;
        if (width == 0) return par_num - 1;
        const int p = col / width;
        return p >= par_num ? par_num - 1 : p;
    };
    // cnt[p*V + i] = edges of row i falling in partition p
    std::vector<int> cnt((size_t)par_num * V, 0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < V; ++i)
        for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
            const int p = part_of(idx[e]);
            if (p >= 0) cnt[(size_t)p * V + i]++;
        }
    // group / edge offsets in (partition-major, row-minor) order
    std::vector<int> goff((size_t)par_num * V + 1), eoff((size_t)par_num * V + 1);
    goff[0] = eoff[0] = 0;
    for (size_t k = 0; k < (size_t)par_num * V; ++k) {
        const int c = cnt[k];
        goff[k + 1] = goff[k] + (ng > 0 ? (c + ng - 1) / ng : (c != 0));
        eoff[k + 1] = eoff[k] + c;
    }
    const int G = goff[(size_t)par_num * V];
    ptr_out[0] = 0;
#pragma omp parallel
    {
    std::vector<int> cursor(par_num);  // per-partition write cursors of the current row
#pragma omp for schedule(dynamic, 1024)
    for (int i = 0; i < V; ++i) {
        std::fill(cursor.begin(), cursor.end(), 0);
        for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
            const int p = part_of(idx[e]);
            if (p < 0) continue;
            const size_t k = (size_t)p * V + i;
            const int pos = eoff[k] + cursor[p]++;
            idx_out[pos] = idx[e];
            if (val && val_out) val_out[pos] = val[e];
            if (eid_out) eid_out[pos] = e;
        }
        for (int p = 0; p < par_num; ++p) {
            const size_t k = (size_t)p * V + i;
            const int c = cnt[k];
            const int ngroups = goff[k + 1] - goff[k];
            for (int q = 0; q < ngroups; ++q) {
                const int g = goff[k] + q;
                const int hi = ng > 0 ? std::min((q + 1) * ng, c) : c;
                ptr_out[g + 1] = eoff[k] + hi;
                target_out[g] = i;
            }
        }
    }
    }
    return G;
}

// ------------------------------------------------------------------------------------- loader
static bool file_exists(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

static bool read_all(const std::string &path, std::vector<char> &buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n + 1);
    size_t got = fread(buf.data(), 1, (size_t)n, f);
    fclose(f);
    buf[got] = 0;
    buf.resize(got + 1);
    return true;
}

// scans `count` whitespace-separated decimal ints starting at *pp; advances *pp.
static bool scan_ints(const char **pp, const char *end, int *out, size_t count)
{
    const char *p = *pp;
    for (size_t k = 0; k < count; ++k) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        if (p >= end) return false;
        bool neg = false;
        if (*p == '-') { neg = true; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        long v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        out[k] = (int)(neg ? -v : v);
    }
    *pp = p;
    return true;
}

static bool read_raw(const std::string &path, int *dst, size_t count)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    size_t got = fread(dst, sizeof(int), count, f);
    fclose(f);
    return got == count;
}

static void write_raw(const std::string &path, const int *src, size_t count)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return;  // cache is best-effort (read-only data dirs are fine)
    fwrite(src, sizeof(int), count, f);
    fclose(f);
}

// reference src/data.cu:31-139 (formats: SURVEY.md appendix).
int load_graph(const char *datadir, const char *dset, const char *suffix, int shuffle, int *num_v, int *num_e,
               int **ptr_o, int **idx_o, int **rows_o, int **rrows_o)
{
    std::string dir = datadir ? datadir : "../data/";
    if (!dir.empty() && dir.back() != '/') dir += '/';
    const std::string base = dir + dset;
    const std::string graph = base + ".graph";
    const std::string ptrfile = graph + ".ptrdump", edgefile = graph + ".edgedump";

    std::vector<char> buf;
    if (!read_all(base + ".config", buf)) return fail(GNNAGG_ERR_IO, "missing config file " + base + ".config");
    int ve[2];
    {
        const char *p = buf.data();
        if (!scan_ints(&p, buf.data() + buf.size() - 1, ve, 2)) return fail(GNNAGG_ERR_IO, "malformed " + base + ".config");
    }
    const int V = ve[0], E = ve[1];
    if (V < 0 || E < 0) return fail(GNNAGG_ERR_IO, "negative sizes in " + base + ".config");
    int *ptr = (int *)malloc(sizeof(int) * ((size_t)V + 1));
    int *idx = (int *)malloc(sizeof(int) * std::max<size_t>((size_t)E, 1));
    auto bail = [&](int code, const std::string &msg) {
        free(ptr);
        free(idx);
        return fail(code, msg);
    };

    const bool have_ptr = file_exists(ptrfile), have_edge = file_exists(edgefile);
    std::vector<char> text;
    const char *cur = nullptr, *end = nullptr;
    if (!have_ptr || !have_edge) {
        if (!read_all(graph, text)) return bail(GNNAGG_ERR_IO, "missing graph file " + graph);
        cur = text.data();
        end = text.data() + text.size() - 1;
    }
    if (have_ptr) {
        if (!read_raw(ptrfile, ptr, (size_t)V + 1)) return bail(GNNAGG_ERR_IO, "short read " + ptrfile);
        if (!have_edge) {  // skip the ptr line of the text file to reach the indices
            std::vector<int> skip((size_t)V + 1);
            if (!scan_ints(&cur, end, skip.data(), (size_t)V + 1)) return bail(GNNAGG_ERR_IO, "malformed " + graph);
        }
    } else {
        if (!scan_ints(&cur, end, ptr, (size_t)V + 1)) return bail(GNNAGG_ERR_IO, "malformed ptr line in " + graph);
        write_raw(ptrfile, ptr, (size_t)V + 1);
    }
    if (ptr[V] != E) return bail(GNNAGG_ERR_IO, "indptr[num_v] != num_e in " + graph);
    if (have_edge) {
        if (!read_raw(edgefile, idx, (size_t)E)) return bail(GNNAGG_ERR_IO, "short read " + edgefile);
    } else {
        if (!scan_ints(&cur, end, idx, (size_t)E)) return bail(GNNAGG_ERR_IO, "malformed idx line in " + graph);
        write_raw(edgefile, idx, (size_t)E);
    }

    int *rows = nullptr, *rrows = nullptr;
    const std::string rfile = base + ".reorder" + (suffix ? suffix : "");
    if (shuffle && suffix && suffix[0] && file_exists(rfile)) {
        std::vector<char> rbuf;
        read_all(rfile, rbuf);
        rows = (int *)malloc(sizeof(int) * std::max<size_t>((size_t)V, 1));
        rrows = (int *)malloc(sizeof(int) * std::max<size_t>((size_t)V, 1));
        const char *p = rbuf.data();
        bool ok = scan_ints(&p, rbuf.data() + rbuf.size() - 1, rows, (size_t)V);
        if (ok) {
            std::vector<char> seen((size_t)V, 0);
            for (int i = 0; i < V && ok; ++i) {
                ok = rows[i] >= 0 && rows[i] < V && !seen[rows[i]];
                if (ok) seen[rows[i]] = 1;
            }
        }
        if (!ok) {
            free(rows);
            free(rrows);
            return bail(GNNAGG_ERR_IO, "reorder file is not a permutation of [0,V): " + rfile);
        }
        for (int i = 0; i < V; ++i) rrows[rows[i]] = i;
        int *nptr = (int *)malloc(sizeof(int) * ((size_t)V + 1));
        int *nidx = (int *)malloc(sizeof(int) * std::max<size_t>((size_t)E, 1));
        reorder_csr(ptr, idx, rows, rrows, V, nptr, nidx);
        free(ptr);
        free(idx);
        ptr = nptr;
        idx = nidx;
    }
    *num_v = V;
    *num_e = E;
    *ptr_o = ptr;
    *idx_o = idx;
    if (rows_o) *rows_o = rows; else free(rows);
    if (rrows_o) *rrows_o = rrows; else free(rrows);
    return GNNAGG_OK;
}

// --------------------------------------------------------------------------- 1-D row partition
// Contiguous row blocks balanced by nnz + rows: cut p is the first row whose prefix weight reaches
// p/nparts of the total (weight of a row = its nnz + 1 so empty-row ranges still spread).
void partition_rows(const int *ptr, int V, int nparts, int *bounds)
{
    const long total = (long)ptr[V] + V;
    bounds[0] = 0;
    int row = 0;
    for (int p = 1; p < nparts; ++p) {
        const long want = total * p / nparts;
        while (row < V && (long)ptr[row] + row < want) ++row;
        bounds[p] = std::max(row, bounds[p - 1]);
    }
    bounds[nparts] = V;
}

// Local CSR of `rank` with columns renumbered into [0, n_local + n_halo): see gnnagg.h Section D.
// Slice form: the rank only holds ITS rows -- ptr_slice[0 .. n_local] (any base offset) and the global column ids of
// those rows -- plus the partition bounds; nothing of the other ranks' rows is read.
int halo_plan_slice(const int *ptr_slice, const int *idx_slice, int num_cols, const int *bounds, int nparts, int rank, int *lptr,
                    int *lidx, int **halo_ids_o, int *halo_counts, int *num_halo)
{
    const int r0 = bounds[rank], r1 = bounds[rank + 1];
    const int nloc = r1 - r0;
    const int base = ptr_slice[0];
    const int ne = ptr_slice[nloc] - base;
    std::vector<int> remote;
    remote.reserve((size_t)ne / 4 + 16);
    for (int e = 0; e < ne; ++e) {
        const int c = idx_slice[e];
        if (c < 0 || c >= num_cols) return fail(GNNAGG_ERR_ARG, "column id out of range in halo_plan");
        if (c < r0 || c >= r1) remote.push_back(c);
    }
    std::sort(remote.begin(), remote.end());
    remote.erase(std::unique(remote.begin(), remote.end()), remote.end());
    // ascending global id == grouped by owner (owners are contiguous id ranges)
    for (int p = 0; p < nparts; ++p) {
        auto lo = std::lower_bound(remote.begin(), remote.end(), bounds[p]);
        auto hi = std::lower_bound(remote.begin(), remote.end(), bounds[p + 1]);
        halo_counts[p] = (int)(hi - lo);
    }
    for (int i = 0; i <= nloc; ++i) lptr[i] = ptr_slice[i] - base;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < ne; ++e) {
        const int c = idx_slice[e];
        if (c >= r0 && c < r1)
            lidx[e] = c - r0;
        else
            lidx[e] = nloc + (int)(std::lower_bound(remote.begin(), remote.end(), c) - remote.begin());
    }
    int *ids = (int *)malloc(sizeof(int) * std::max<size_t>(remote.size(), 1));
    std::copy(remote.begin(), remote.end(), ids);
    *halo_ids_o = ids;
    *num_halo = (int)remote.size();
    return GNNAGG_OK;
}

// ---- staged exchange: how one (reader <- owner) list of n rows is cut into stages.  Both ends of a pair call this with the same n,
// so they cut the list the same way.  mode 0 ("stripe"): slice j of k of every list; mode 1 ("owner"): the whole list in the stage
// of the pair's ring distance (reader - owner) mod world - 1.
static long long stage_share(long long n, int reader, int owner, int world, int mode, int k, int stage)
{
    if (mode == 1) return ((reader - owner - 1 + 2 * world) % world) == stage ? n : 0;
    return n * (stage + 1) / k - n * stage / k;
}

int halo_stage_count(int world, int mode, int k) { return world <= 1 ? 1 : mode == 1 ? std::max(world - 1, 1) : std::max(k, 1); }

// Receiving side: recv_rows[owner] rows in owner-major order -> stage_recv[S][world] and, for every owner-major halo slot, its
// stage-major slot (stage, then owner, then the list's own order).
void halo_stage_plan_recv(const long long *recv_rows, int world, int rank, int mode, int k, long long *stage_recv, int *new_of_old)
{
    const int S = halo_stage_count(world, mode, k);
    std::vector<long long> owner0((size_t)world + 1, 0), taken((size_t)world, 0);
    for (int o = 0; o < world; ++o) owner0[o + 1] = owner0[o] + recv_rows[o];
    long long pos = 0;
    for (int s = 0; s < S; ++s)
        for (int o = 0; o < world; ++o) {
            const long long c = (world <= 1 || S == 1) ? recv_rows[o] : stage_share(recv_rows[o], rank, o, world, mode, k, s);
            stage_recv[(size_t)s * world + o] = c;
            if (new_of_old)
                for (long long i = 0; i < c; ++i) new_of_old[owner0[o] + taken[o] + i] = (int)(pos + i);
            taken[o] += c;
            pos += c;
        }
}

// Sending side: send_rows[reader] rows in reader-major order (every reader's list in ITS arrival order: stage by stage) ->
// stage_send[S][world] and order[stage-major position] = index into the reader-major list.
void halo_stage_plan_send(const long long *send_rows, int world, int rank, int mode, int k, long long *stage_send, int *order)
{
    const int S = halo_stage_count(world, mode, k);
    std::vector<long long> reader0((size_t)world + 1, 0), taken((size_t)world, 0);
    for (int q = 0; q < world; ++q) reader0[q + 1] = reader0[q] + send_rows[q];
    long long pos = 0;
    for (int s = 0; s < S; ++s)
        for (int q = 0; q < world; ++q) {
            const long long c = (world <= 1 || S == 1) ? send_rows[q] : stage_share(send_rows[q], q, rank, world, mode, k, s);
            stage_send[(size_t)s * world + q] = c;
            if (order)
                for (long long i = 0; i < c; ++i) order[pos + i] = (int)(reader0[q] + taken[q] + i);
            taken[q] += c;
            pos += c;
        }
}

int halo_plan(const int *ptr, const int *idx, int V, const int *bounds, int nparts, int rank, int *lptr,
              int *lidx, int **halo_ids_o, int *halo_counts, int *num_halo)
{
    const int r0 = bounds[rank];
    return halo_plan_slice(ptr + r0, idx + ptr[r0], V, bounds, nparts, rank, lptr, lidx, halo_ids_o, halo_counts, num_halo);
}

}  // namespace gnnagg

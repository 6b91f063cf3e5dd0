// Sanitizer harness for the host-side code of the library (host_graph.cpp, reorder.cpp): built with g++ -fsanitize=thread or
// -fsanitize=address,undefined by tests/test_host_sanitizers.py (CPU only; GPU sanitizers are not available on the pool).
// Exercises the schedulers, the partitioner / halo plan and the reorder generator -- serial and with several walkers -- on a
// seeded power-law graph and checks what the Python tests check (valid permutation, group arrays that cover every edge once).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../gnn_computing_amd/csrc/common.h"

namespace gnnagg {
int fail(int code, const std::string &msg)
{
    fprintf(stderr, "fail(%d): %s\n", code, msg.c_str());
    return code;
}
}  // namespace gnnagg

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char **argv)
{
    const int V = argc > 1 ? atoi(argv[1]) : 6000, E = argc > 2 ? atoi(argv[2]) : 120000;
    // power-law-ish degrees, half of the sources near the row (so that the reorder generator has clusters to find)
    std::vector<int> deg(V, 0);
    for (int e = 0; e < E; ++e) {
        const uint32_t r = rnd();
        deg[(int)((uint64_t)(r % 1000u) * (r % 1000u) * V / 1000000u) % V]++;
    }
    deg[V / 2] = 0;
    std::vector<int> ptr(V + 1, 0);
    for (int v = 0; v < V; ++v) ptr[v + 1] = ptr[v] + deg[v];
    std::vector<int> idx(ptr[V]);
    for (int v = 0; v < V; ++v)
        for (int k = ptr[v]; k < ptr[v + 1]; ++k) idx[k] = (rnd() & 1) ? (int)(rnd() % (uint32_t)V) : (int)((v + rnd() % 64u) % (uint32_t)V);
    const int NE = ptr[V];
    int rc = 0;
    // schedulers
    {
        const int G = gnnagg::neighbor_grouping(ptr.data(), 16, V, nullptr, nullptr);
        std::vector<int> ps(G + 1), tg(G);
        gnnagg::neighbor_grouping(ptr.data(), 16, V, ps.data(), tg.data());
        if (ps[G] != NE) { fprintf(stderr, "neighbor grouping does not cover the edges\n"); return 2; }
        std::vector<float> val(NE, 1.0f), lv(NE);
        for (int ng : {0, 8}) {
            std::vector<int> lp(NE + (size_t)V * 7 + 2), li(NE), lt(NE + (size_t)V * 7 + 1), eid(NE);
            const int LG = gnnagg::locality_schedule(ptr.data(), idx.data(), val.data(), 7, ng, V, V, lp.data(), li.data(), lv.data(), lt.data(), eid.data());
            if (LG < 0 || lp[LG] != NE) { fprintf(stderr, "locality schedule does not cover the edges\n"); return 2; }
        }
    }
    // partition + halo plan
    {
        const int parts = 3;
        std::vector<int> bounds(parts + 1);
        gnnagg::partition_rows(ptr.data(), V, parts, bounds.data());
        for (int r = 0; r < parts; ++r) {
            const int nl = bounds[r + 1] - bounds[r];
            std::vector<int> lptr(nl + 1), lidx(ptr[bounds[r + 1]] - ptr[bounds[r]] + 1), counts(parts);
            int *halo = nullptr, nh = 0;
            rc |= gnnagg::halo_plan(ptr.data(), idx.data(), V, bounds.data(), parts, r, lptr.data(), lidx.data(), &halo, counts.data(), &nh);
            free(halo);
        }
    }
    // reorder generator: reference order, serial greedy, several walkers (GNNAGG_REORDER_WALKERS is read per call)
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 2) setenv("GNNAGG_REORDER_WALKERS", "4", 1); else setenv("GNNAGG_REORDER_WALKERS", "1", 1);
        std::vector<int> rows(V, -1);
        int nc = 0;
        rc |= gnnagg::cluster_reorder(ptr.data(), idx.data(), V, 0.2, 64, pass == 0 ? 64 : 1, 123, 8, rows.data(), &nc, pass == 0 ? 0 : 1, 512);
        std::vector<char> seen(V, 0);
        for (int v : rows) {
            if (v < 0 || v >= V || seen[v]) { fprintf(stderr, "pass %d: not a permutation\n", pass); return 3; }
            seen[v] = 1;
        }
        std::vector<int> rev(V), np(V + 1), ni(NE);
        for (int i = 0; i < V; ++i) rev[rows[i]] = i;
        gnnagg::reorder_csr(ptr.data(), idx.data(), rows.data(), rev.data(), V, np.data(), ni.data());
        if (np[V] != NE) { fprintf(stderr, "reorder_csr lost edges\n"); return 4; }
    }
    printf("host sanitize harness: ok (V %d, E %d, rc %d)\n", V, NE, rc);
    return rc ? 1 : 0;
}

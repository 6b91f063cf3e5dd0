// api_extras.hip -- entry points only libgnnagg_extras.so has (GNNAGG_EXTRAS; `make -C gnn_computing_amd/csrc extras`): the backward
// passes.  The reference is forward-only; its one backward kernel is the "Experiment" block of include/aggr_gat.h:222-296 (no caller),
// which SURVEY 2.2 marks out of scope -- built in rounds 2-3 as the mathematics its comments describe, kept for
// gnn_computing_amd/extras/autograd.py and the second-tier tests, not shipped in the default library.
#include "api_internal.h"

#ifdef GNNAGG_EXTRAS
using namespace gnnagg;

#pragma GCC visibility push(default)
extern "C" {

static int build_transposed(Ctx *c)
{
    if (c->tr.valid) return GNNAGG_OK;
    int rc = fetch_host_ptr(c);
    if (rc) return rc;
    const int V = c->V, E = c->E;
    std::vector<int> h_idx((size_t)E);
    if (int rc_ = copy_to_host(c, h_idx.data(), c->d_idx, (size_t)E * sizeof(int))) return rc_;
    std::vector<int> ptr_t((size_t)V + 1, 0), idx_t((size_t)E), perm((size_t)E);
    for (int e = 0; e < E; ++e) {
        if (h_idx[e] < 0 || h_idx[e] >= V) return fail(GNNAGG_ERR_ARG, "run_bwd: neighbor id outside [0, num_v)");
        ++ptr_t[(size_t)h_idx[e] + 1];
    }
    for (int v = 0; v < V; ++v) ptr_t[v + 1] += ptr_t[v];
    std::vector<int> fill(ptr_t.begin(), ptr_t.end() - 1);
    for (int r = 0; r < V; ++r)
        for (int e = c->h_ptr[r]; e < c->h_ptr[r + 1]; ++e) {  // counting sort: stable in the original edge order
            const int pos = fill[h_idx[e]]++;
            idx_t[pos] = r;
            perm[pos] = e;
        }
    Ctx::Transposed &t = c->tr;
    if ((rc = t.ptr_t.upload(ptr_t)) || (rc = t.idx_t.upload(idx_t)) || (rc = t.perm.upload(perm))) return rc;
    if ((rc = t.val_t.reserve((size_t)std::max(E, 1))) || (rc = t.dz.reserve((size_t)std::max(E, 1))) ||
        (rc = t.dz_t.reserve((size_t)std::max(E, 1))) || (rc = t.rowdot.reserve((size_t)std::max(V, 1))) ||
        (rc = t.da.reserve((size_t)std::max(V, 1))) || (rc = t.db.reserve((size_t)std::max(V, 1))))
        return rc;
    if ((rc = gnnagg_gcn_create(t.ptr_t.p, t.idx_t.p, t.val_t.p, V, E, &t.agg))) return rc;
    t.valid = true;
    return GNNAGG_OK;
}

int gnnagg_gat_run_bwd(gnnagg_handle h, const float *d_output, const float *d_doutput, const float *d_newval, const float *d_div,
                       const float *d_infeat, float *d_a_b_grad, float *d_feat_grad, float relu_slope, int feat)
{
    GET_CTX(h);
    if (c->kind != Ctx::GAT) return fail(GNNAGG_ERR_ARG, "handle is not a GAT aggregator");
    if (feat <= 0 || !d_output || !d_doutput || !d_div || !d_infeat || !d_a_b_grad || !d_feat_grad || (!d_newval && c->E > 0))
        return fail(GNNAGG_ERR_ARG, "bad run_bwd arguments");
    int rc = build_transposed(c);
    if (rc) return rc;
    Ctx::Transposed &t = c->tr;
    Ctx *ct = lookup(t.agg);
    if (!ct) return fail(GNNAGG_ERR_ARG, "run_bwd: transposed aggregator lost");
    ct->stream = c->stream;
    // 1. per-edge dz on the chunked work items of this graph
    EdgeItemLaunch L;
    if ((rc = edge_launch(c, L, 1))) return rc;
    GatBwdLaunch B;
    B.wl = L.wl; B.idx = c->d_idx; B.out = d_output; B.dout = d_doutput; B.newval = d_newval; B.div = d_div; B.x = d_infeat;
    B.rowdot = t.rowdot.p; B.dz = t.dz.p; B.V = c->V; B.feat = feat; B.slope = relu_slope;
    if ((rc = launch_gat_bwd_edges(B, c->stream))) return rc;
    // 2. centre-term gradient: row sums of dz (the hub-safe, ordered add_to_center)
    L.in = t.dz.p; L.den = t.da.p;
    if ((rc = launch_edge_items_sum(L, 1, c->stream))) return rc;
    // 3. the source side runs on the transposed graph
    if ((rc = launch_gat_bwd_permute(t.perm.p, t.idx_t.p, t.dz.p, d_newval, d_div, t.dz_t.p, t.val_t.p, c->E, c->stream))) return rc;
    EdgeItemLaunch LT;
    if ((rc = edge_launch(ct, LT, 1))) return rc;
    LT.in = t.dz_t.p; LT.den = t.db.p;
    if ((rc = launch_edge_items_sum(LT, 1, c->stream))) return rc;
    if ((rc = launch_interleave2(t.da.p, t.db.p, d_a_b_grad, c->V, c->stream))) return rc;
    // 4. d_feat = A^T-aggregation of dout with edge values p (balanced GCN kernels)
    return gcn_run(ct, d_doutput, d_feat_grad, feat, GNNAGG_MODE_BALANCED, GNNAGG_REDUCE_SUM);
}

int gnnagg_gcn_run_bwd(gnnagg_handle h, const float *d_doutput, float *d_dinput, int feat)
{
    GET_CTX(h);
    if (c->kind != Ctx::GCN) return fail(GNNAGG_ERR_ARG, "handle is not a GCN aggregator");
    if (feat <= 0 || !d_doutput || !d_dinput) return fail(GNNAGG_ERR_ARG, "bad gcn_run_bwd arguments");
    int rc = build_transposed(c);
    if (rc) return rc;
    Ctx::Transposed &t = c->tr;
    Ctx *ct = lookup(t.agg);
    if (!ct) return fail(GNNAGG_ERR_ARG, "gcn_run_bwd: transposed aggregator lost");
    ct->stream = c->stream;
    // the edge values follow their edges (re-gathered every call: updateval may have re-aliased them)
    if ((rc = launch_permute_val(t.perm.p, c->d_val, t.val_t.p, c->E, c->stream))) return rc;
    return gcn_run(ct, d_doutput, d_dinput, feat, GNNAGG_MODE_BALANCED, GNNAGG_REDUCE_SUM);
}

}  // extern "C"
#pragma GCC visibility pop
#endif  // GNNAGG_EXTRAS

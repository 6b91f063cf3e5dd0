// api_flat.hip -- the parts of the C-ABI that hold no state of their own (include/gnnagg.h):
//   Section A  the flat API of the reference's torch binding, name for name (Figure7/kernel.cpp:15-35, kernel_generated.cu:15-74): thin
//              wrappers over Section B that abort like the reference (util.h:82-104) unless gnnagg_set_abort_on_error(0);
//   Section C  host graph preparation (load_graph, reorderCSR, the schedulers, the reorder generator: host_graph.cpp, reorder.cpp);
//   Section D  row-partition planning and the pack kernel's entry point (the RCCL transport itself is dist_rccl.cpp).
#include "api_internal.h"

using namespace gnnagg;

#pragma GCC visibility push(default)
extern "C" {

// ------------------------------------------------------------------------------- Section A
int64_t GCN_init_impl(int *ptr, int *idx, float *val, int num_v, int num_e)
{
    gnnagg_handle h = 0;
    die_if_abort(gnnagg_gcn_create(ptr, idx, val, num_v, num_e, &h), "GCN_init_impl");
    if (h) (void)gnnagg_set_option(h, "reference_defaults", 1);
    return h;
}

void GCN_update_val_impl(int64_t at, float *val) { die_if_abort(gnnagg_update_val(at, val), "GCN_update_val_impl"); }

void GCN_run_impl(int64_t at, float *feat, float *out_feat, int blocksize, int scheduled, int featlen)
{
    (void)blocksize;
    die_if_abort(gnnagg_gcn_run(at, feat, out_feat, featlen, scheduled ? GNNAGG_MODE_SCHEDULED : GNNAGG_MODE_ROWS,
                                GNNAGG_REDUCE_SUM),
                 "GCN_run_impl");
}

static int schedule_ng(int64_t at, int *arr)
{
    GET_CTX(at);
    return do_schedule(c, GNNAGG_SCHED_NEIGHBOR_GROUPING, arr, c->V);
}

void GCN_schedule_impl(int64_t at, int *arr) { die_if_abort(schedule_ng(at, arr), "GCN_schedule_impl"); }

int64_t GAT_init_impl(int *ptr, int *idx, int num_v, int num_e)
{
    gnnagg_handle h = 0;
    die_if_abort(gnnagg_gat_create(ptr, idx, num_v, num_e, &h), "GAT_init_impl");
    if (h) (void)gnnagg_set_option(h, "reference_defaults", 1);
    return h;
}

void GAT_run_impl(int64_t at, float *feat, float *att, float *out_feat, int blocksize, int scheduled, int featlen)
{
    (void)blocksize;
    die_if_abort(gnnagg_gat_run(at, feat, att, out_feat, featlen, 1, 0.2f,
                                scheduled ? GNNAGG_MODE_SCHEDULED : GNNAGG_MODE_ROWS, nullptr),
                 "GAT_run_impl");
}

void GAT_run_u_add_v_impl(int64_t at, float *att, float *outval, int blocksize)
{
    (void)blocksize;
    die_if_abort(gnnagg_gat_run_u_add_v(at, att, outval), "GAT_run_u_add_v_impl");
}

void GAT_run_add_to_center_impl(int64_t at, float *inval, float *outatt, int blocksize)
{
    (void)blocksize;
    die_if_abort(gnnagg_gat_run_add_to_center(at, inval, outatt), "GAT_run_add_to_center_impl");
}

void GAT_run_div_each_impl(int64_t at, float *inatt, float *inoutval, int blocksize)
{
    (void)blocksize;
    die_if_abort(gnnagg_gat_run_div_each(at, inatt, inoutval), "GAT_run_div_each_impl");
}

void GAT_schedule_impl(int64_t at, int *arr) { die_if_abort(schedule_ng(at, arr), "GAT_schedule_impl"); }

// ------------------------------------------------------------------------------- Section C
int gnnagg_load_graph(const char *datadir, const char *dset, const char *reorder_suffix, int shuffle, int *num_v,
                      int *num_e, int **h_ptr, int **h_idx, int **h_rows, int **h_reverse_rows)
{
    if (!dset || !num_v || !num_e || !h_ptr || !h_idx) return fail(GNNAGG_ERR_ARG, "bad load_graph arguments");
    if (h_rows) *h_rows = nullptr;
    if (h_reverse_rows) *h_reverse_rows = nullptr;
    return load_graph(datadir, dset, reorder_suffix, shuffle, num_v, num_e, h_ptr, h_idx, h_rows, h_reverse_rows);
}

void gnnagg_free_host(void *p) { free(p); }

int gnnagg_reorder_csr(const int *h_ptr, const int *h_idx, const int *h_map, const int *h_reverse_map, int num_v,
                       int num_e, int *h_newptr, int *h_newidx)
{
    if (!h_ptr || !h_map || !h_reverse_map || !h_newptr || num_v < 0 || num_e < 0 || (num_e > 0 && (!h_idx || !h_newidx)))
        return fail(GNNAGG_ERR_ARG, "bad reorder_csr arguments");
    reorder_csr(h_ptr, h_idx, h_map, h_reverse_map, num_v, h_newptr, h_newidx);
    return GNNAGG_OK;
}

int gnnagg_neighbor_grouping_schedule(const int *h_ptr, int neighbor_num, int num_v, int *h_ptr_out,
                                      int *h_target_out, int *num_groups)
{
    if (!h_ptr || neighbor_num <= 0 || num_v < 0 || !num_groups)
        return fail(GNNAGG_ERR_ARG, "bad neighbor_grouping_schedule arguments");
    *num_groups = neighbor_grouping(h_ptr, neighbor_num, num_v, h_ptr_out, h_target_out);
    return GNNAGG_OK;
}

int gnnagg_locality_schedule(const int *h_ptr, const int *h_idx, const float *h_val, int par_num, int neighbor_num,
                             int num_v, int total_num_v, int *h_ptr_out, int *h_idx_out, float *h_val_out,
                             int *h_target_out, int *num_groups)
{
    if (!h_ptr || !h_idx || par_num <= 0 || num_v < 0 || !h_ptr_out || !h_idx_out || !h_target_out || !num_groups)
        return fail(GNNAGG_ERR_ARG, "bad locality_schedule arguments");
    *num_groups = locality_schedule(h_ptr, h_idx, h_val, par_num, neighbor_num, num_v, total_num_v, h_ptr_out,
                                    h_idx_out, h_val_out, h_target_out);
    return GNNAGG_OK;
}

int gnnagg_cluster_reorder(const int *h_ptr, const int *h_idx, int num_v, float threshold, int num_perm, int cluster_cap,
                           unsigned long long seed, int *h_rows_out, int *num_clusters)
{
    if (!h_ptr || !h_rows_out || num_v < 0 || (h_ptr[num_v] > 0 && !h_idx))
        return fail(GNNAGG_ERR_ARG, "bad cluster_reorder arguments");
    return cluster_reorder(h_ptr, h_idx, num_v, threshold > 0 ? threshold : 0.2, num_perm > 0 ? num_perm : 64,
                           cluster_cap > 0 ? cluster_cap : 64, seed, 8, h_rows_out, num_clusters);
}

int gnnagg_cluster_reorder_ex(const int *h_ptr, const int *h_idx, int num_v, float threshold, int num_perm, int cluster_cap,
                              unsigned long long seed, int order_mode, int cache_rows, int *h_rows_out, int *num_clusters)
{
    if (!h_ptr || !h_rows_out || num_v < 0 || (h_ptr[num_v] > 0 && !h_idx))
        return fail(GNNAGG_ERR_ARG, "bad cluster_reorder arguments");
    return cluster_reorder(h_ptr, h_idx, num_v, threshold > 0 ? threshold : 0.2, num_perm > 0 ? num_perm : 64,
                           cluster_cap > 0 ? cluster_cap : 64, seed, 8, h_rows_out, num_clusters, order_mode,
                           cache_rows > 0 ? cache_rows : 4096);
}

// ------------------------------------------------------------------------------- Section D
int gnnagg_partition_rows(const int *h_ptr, int num_v, int nparts, int *h_bounds)
{
    if (!h_ptr || !h_bounds || nparts <= 0 || num_v < 0) return fail(GNNAGG_ERR_ARG, "bad partition_rows arguments");
    partition_rows(h_ptr, num_v, nparts, h_bounds);
    return GNNAGG_OK;
}

int gnnagg_halo_plan(const int *h_ptr, const int *h_idx, int num_v, const int *h_bounds, int nparts, int rank,
                     int *h_local_ptr, int *h_local_idx, int **h_halo_ids, int *h_halo_counts, int *num_halo)
{
    if (!h_ptr || !h_idx || !h_bounds || nparts <= 0 || rank < 0 || rank >= nparts || !h_local_ptr || !h_local_idx ||
        !h_halo_ids || !h_halo_counts || !num_halo)
        return fail(GNNAGG_ERR_ARG, "bad halo_plan arguments");
    return halo_plan(h_ptr, h_idx, num_v, h_bounds, nparts, rank, h_local_ptr, h_local_idx, h_halo_ids, h_halo_counts,
                     num_halo);
}

int gnnagg_halo_plan_slice(const int *h_ptr_slice, const int *h_idx_slice, int num_cols, const int *h_bounds, int nparts, int rank,
                           int *h_local_ptr, int *h_local_idx, int **h_halo_ids, int *h_halo_counts, int *num_halo)
{
    if (!h_ptr_slice || !h_idx_slice || !h_bounds || nparts <= 0 || rank < 0 || rank >= nparts || num_cols < 0 || !h_local_ptr ||
        !h_local_idx || !h_halo_ids || !h_halo_counts || !num_halo)
        return fail(GNNAGG_ERR_ARG, "bad halo_plan_slice arguments");
    return halo_plan_slice(h_ptr_slice, h_idx_slice, num_cols, h_bounds, nparts, rank, h_local_ptr, h_local_idx, h_halo_ids,
                           h_halo_counts, num_halo);
}

int gnnagg_halo_stage_plan(const long long *h_recv_rows, const long long *h_send_rows, int world, int rank, int mode, int k, int *n_stages,
                           long long *h_stage_recv, int *h_new_of_old, long long *h_stage_send, int *h_send_order)
{
    if (world <= 0 || rank < 0 || rank >= world || (mode != GNNAGG_STAGES_STRIPE && mode != GNNAGG_STAGES_OWNER) || (mode == GNNAGG_STAGES_STRIPE && (k < 1 || k > 64)) ||
        (mode == GNNAGG_STAGES_OWNER && world > 65) || !n_stages)
        return fail(GNNAGG_ERR_ARG, "bad halo_stage_plan arguments (stripe: 1 <= k <= 64; owner: world <= 65)");
    *n_stages = halo_stage_count(world, mode, k);
    if (h_recv_rows) {
        if (!h_stage_recv) return fail(GNNAGG_ERR_ARG, "halo_stage_plan: null stage_recv");
        halo_stage_plan_recv(h_recv_rows, world, rank, mode, k, h_stage_recv, h_new_of_old);
    }
    if (h_send_rows) {
        if (!h_stage_send) return fail(GNNAGG_ERR_ARG, "halo_stage_plan: null stage_send");
        halo_stage_plan_send(h_send_rows, world, rank, mode, k, h_stage_send, h_send_order);
    }
    return GNNAGG_OK;
}

int gnnagg_pack_rows(const float *d_x, const int *d_ids, int n, int feat, float *d_out, void *hip_stream)
{
    if (n < 0 || feat <= 0 || (n > 0 && (!d_x || !d_ids || !d_out))) return fail(GNNAGG_ERR_ARG, "bad pack_rows arguments");
    return launch_pack_rows(d_x, d_ids, n, feat, d_out, hip_stream);
}

}  // extern "C"
#pragma GCC visibility pop

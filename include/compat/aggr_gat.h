// aggr_gat.h -- class Aggregator_GAT with the reference's public surface (reference include/aggr_gat.h:299-441).
#ifndef GNNAGG_COMPAT_AGGR_GAT_H
#define GNNAGG_COMPAT_AGGR_GAT_H
#include "aggregator.h"

class Aggregator_GAT : public Aggregator
{
public:
    // reference aggr_gat.h:302
    Aggregator_GAT(int *host_out_ptr, int *host_out_idx, int *dev_out_ptr, int *dev_out_idx, int out_num_v, int out_num_e,
                   int out_feat_in, int out_feat_out)
        : Aggregator(host_out_ptr, host_out_idx, dev_out_ptr, dev_out_idx, out_num_v, out_num_e, out_feat_in, out_feat_out)
    {
        checkGnnagg(gnnagg_gat_create(d_ptr, d_idx, num_v, num_e, &handle));
        checkGnnagg(gnnagg_set_option(handle, "reference_defaults", 1));  // run(..., scheduled = 0) takes the balanced order
    }
    // reference aggr_gat.h:308
    Aggregator_GAT(CSRSubGraph g, int out_feat_in, int out_feat_out) : Aggregator(g, out_feat_in, out_feat_out)
    {
        checkGnnagg(gnnagg_gat_create(d_ptr, d_idx, num_v, num_e, &handle));
        checkGnnagg(gnnagg_set_option(handle, "reference_defaults", 1));  // run(..., scheduled = 0) takes the balanced order
    }
    // two aggregators may share one CSR (Figure10/main_a.cu:66-70): only one of them may own it
    void releaseGraphOwnership() { d_ptr = nullptr; d_idx = nullptr; d_vset = nullptr; }

    // reference aggr_gat.h:317-354; leaky slope 0.2 (:347); att is [V,2]
    double run(float *vin, float *vatt, float *vout, int BLOCK_SIZE, bool scheduled) override
    {
        return run_with_feat(vin, vatt, vout, BLOCK_SIZE, scheduled, feat_in);
    }
    // reference aggr_gat.h:355-394
    double run_with_feat(float *vin, float *vatt, float *vout, int BLOCK_SIZE, bool scheduled, int feat)
    {
        (void)BLOCK_SIZE;
        feat_in = feat;
        checkGnnagg(gnnagg_gat_run(handle, vin, vatt, vout, feat, 1, 0.2f,
                                   scheduled ? GNNAGG_MODE_SCHEDULED : GNNAGG_MODE_ROWS, nullptr));
        dump_run(scheduled ? "gat_run_s1" : "gat_run_s0", vin, vatt, vout, feat, 1);
        return 0.0;
    }
    // multi-head extension: att [V,heads,2], feat % heads == 0
    double run_heads(float *vin, float *vatt, float *vout, int feat, int heads, int mode = GNNAGG_MODE_BALANCED,
                     float slope = 0.2f)
    {
        checkGnnagg(gnnagg_gat_run(handle, vin, vatt, vout, feat, heads, slope, mode, nullptr));
        dump_run("gat_run_heads", vin, vatt, vout, feat, heads);
        return 0.0;
    }
    // reference aggr_gat.h:395-425
    void run_att(float *in_att, float *out_val, int BLOCK_SIZE)
    {
        (void)BLOCK_SIZE;
        checkGnnagg(gnnagg_gat_run_att(handle, in_att, out_val, 1, 0.2f));
        dump_edge("gat_run_att", in_att, 2, out_val);
    }
    void run_u_add_v(float *in_att, float *out_val, int BLOCK_SIZE)
    {
        (void)BLOCK_SIZE;
        checkGnnagg(gnnagg_gat_run_u_add_v(handle, in_att, out_val));
        dump_edge("gat_run_u_add_v", in_att, 2, out_val);
    }
    void run_add_to_center(float *in_val, float *out_att, int BLOCK_SIZE)
    {
        (void)BLOCK_SIZE;
        if (compat_dump_dir()) compat_dump("gat_run_add_to_center", "val_in", in_val, sizeof(float) * (size_t)num_e);
        checkGnnagg(gnnagg_gat_run_add_to_center(handle, in_val, out_att));
        dump_edge("gat_run_add_to_center", out_att, 1, nullptr);
    }
    void run_div_each(float *in_att, float *in_out_val, int BLOCK_SIZE)
    {
        (void)BLOCK_SIZE;
        if (compat_dump_dir()) compat_dump("gat_run_div_each", "val_in", in_out_val, sizeof(float) * (size_t)num_e);
        checkGnnagg(gnnagg_gat_run_div_each(handle, in_att, in_out_val));
        dump_edge("gat_run_div_each", in_att, 1, in_out_val);
    }
    // aggr_gat.h:426-434.  Same argument order as the reference; d_a_b / d_feat are overwritten (see gnnagg.h).
    void run_bwd(float *output, float *doutput, float *newval, float *div, float *infeat, float *d_a_b, float *d_feat,
                 float relu_l, int BLOCK_SIZE)
    {
        (void)BLOCK_SIZE;
#ifdef GNNAGG_EXTRAS
        checkGnnagg(gnnagg_gat_run_bwd(handle, output, doutput, newval, div, infeat, d_a_b, d_feat, relu_l, feat_in));
#else
        (void)output; (void)doutput; (void)newval; (void)div; (void)infeat; (void)d_a_b; (void)d_feat; (void)relu_l;
        FatalError("run_bwd (aggr_gat.h:426-434, \"Experiment\", no caller in the reference) needs libgnnagg_extras.so and -DGNNAGG_EXTRAS");
#endif
    }

private:
    // GNNAGG_COMPAT_DUMP (util.h): operands of the last call; att is [V, heads, 2] (node-wise terms), val [E] (edge-wise values)
    void dump_run(const char *entry, const float *vin, const float *vatt, const float *vout, int feat, int heads) const
    {
        if (!compat_dump_dir()) return;
        dump_graph(entry);
        compat_dump(entry, "x", vin, sizeof(float) * (size_t)num_v * feat);
        compat_dump(entry, "att", vatt, sizeof(float) * (size_t)num_v * heads * 2);
        compat_dump(entry, "y", vout, sizeof(float) * (size_t)num_v * feat);
    }
    void dump_edge(const char *entry, const float *node_terms, int per_node, const float *edge_vals) const
    {
        if (!compat_dump_dir()) return;
        dump_graph(entry);
        compat_dump(entry, "att", node_terms, sizeof(float) * (size_t)num_v * per_node);
        compat_dump(entry, "val", edge_vals, sizeof(float) * (size_t)num_e);
    }
};
#endif

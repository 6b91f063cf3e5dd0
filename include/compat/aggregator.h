// aggregator.h -- class Aggregator with the reference's public surface (reference include/aggregator.h:25-151)
// as a thin C++ shim over the C-ABI handle of libgnnagg.so.  All compute is in the library's HIP kernels.
//
// Ownership follows the reference: the aggregator takes ownership of the device CSR pointers it is given
// and frees them in its destructor unless they were registered with registerPtr (util.h:154-177).
#ifndef GNNAGG_COMPAT_AGGREGATOR_H
#define GNNAGG_COMPAT_AGGREGATOR_H
#include "data.h"
#include "graph_schedule.h"
#include "util.h"

class Aggregator
{
public:
    // reference aggregator.h:28 (host mirrors are fetched by the library when a schedule needs them)
    Aggregator(int *host_out_ptr, int *host_out_idx, int *dev_out_ptr, int *dev_out_idx, int out_num_v, int out_num_e,
               int out_feat_in, int out_feat_out)
        : feat_in(out_feat_in), feat_out(out_feat_out), d_ptr(dev_out_ptr), d_idx(dev_out_idx), num_v(out_num_v),
          num_e(out_num_e)
    {
        (void)host_out_ptr;
        (void)host_out_idx;
    }
    // reference aggregator.h:42
    Aggregator(CSRSubGraph g, int out_feat_in, int out_feat_out)
        : feat_in(out_feat_in), feat_out(out_feat_out), d_ptr(g.ptr), d_idx(g.idx), d_vset(g.vertexset), num_v(g.num_v),
          num_e(g.num_e) {}
    virtual ~Aggregator()  // (the reference's destructor is not virtual, aggregator.h:58; deleting through a base pointer leaks there)
    {
        if (handle) gnnagg_destroy(handle);
        safeFree(d_ptr);
        safeFree(d_idx);
        safeFree(d_vset);
        safeFree(d_edgelist);
    }
    // reference aggregator.h:67-99; param[0] = NG or par_num, param[1] = NG of the combined schedule
    virtual void schedule(Schedule s, int *param)
    {
        sche = s;
        checkGnnagg(gnnagg_schedule(handle, (int)s, param, n > 0 ? n : num_v));
        checkGnnagg(gnnagg_num_target(handle, GNNAGG_MODE_SCHEDULED, &num_target));
        dbg(num_target);
    }
    virtual double run(float *, float *, int, bool) { assert(false); return -1; }
    virtual double run(float *, float *, float *, int, bool) { assert(false); return -1; }
    virtual double runEdgeWise(float *, float *, int, bool) { assert(false); return -1; }
    // reference aggregator.h:115-122
    void csr2edgelist()
    {
        safeFree(d_edgelist);
        checkHipErrors(hipMalloc2((void **)&d_edgelist, 2 * (size_t)num_e * sizeof(int)));
        checkGnnagg(gnnagg_csr2edgelist(handle, d_edgelist));
    }
    int *edgelist() const { return d_edgelist; }
    // GNNAGG_COMPAT_DUMP: the CSR this aggregator runs on (after load_graph's reorder, if any)
    void dump_graph(const char *entry) const
    {
        if (!compat_dump_dir()) return;
        compat_dump(entry, "ptr", d_ptr, sizeof(int) * ((size_t)num_v + 1));
        compat_dump(entry, "idx", d_idx, sizeof(int) * (size_t)num_e);
    }

    int feat_in = 0;
    int feat_out = 0;
    int num_target = 0;

protected:
    gnnagg_handle handle = 0;
    int *d_ptr = nullptr;
    int *d_idx = nullptr;
    int *d_vset = nullptr;
    int *d_edgelist = nullptr;
    int num_v = 0;
    int num_e = 0;
    Schedule sche = nop;
};
#endif

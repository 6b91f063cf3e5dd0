"""torch.autograd wrappers over the aggregators (extension: the reference is forward-only; its only backward kernel is the
"Experiment" block of include/aggr_gat.h:222-296, which Aggregator_GAT.run_bwd completes).

    y = gcn_aggregate(agg, x)              # y = A.x (sum, the aggregator's edge values), dL/dx = A^T.dL/dy
    y = gat_aggregate(gat, x, att)         # single-head fused GAT, att [V,2]; gradients w.r.t. x and att

Both run the HIP kernels in forward and backward; nothing here falls back to torch ops."""
import torch


class _GcnAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, agg, mode):
        x = x.contiguous()
        y = torch.empty((agg.num_v, x.shape[1]), dtype=torch.float32, device=x.device)
        agg.run(x, y, 128, mode)
        ctx.agg = agg
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ctx.agg.run_bwd(dy, dx)
        return dx, None, None


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, att, gat, mode, slope):
        x, att = x.contiguous(), att.contiguous()
        V, F = gat.num_v, x.shape[1]
        y = torch.empty((V, F), dtype=torch.float32, device=x.device)
        newval = torch.empty((gat.num_e, 1), dtype=torch.float32, device=x.device)
        gat.run(x, att, y, 128, mode, heads=1, slope=slope, newval=newval)
        div = torch.empty(V, dtype=torch.float32, device=x.device)
        gat.run_add_to_center(newval, div)
        ctx.gat, ctx.slope = gat, slope
        ctx.save_for_backward(x, y, newval, div)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, newval, div = ctx.saved_tensors
        dy = dy.contiguous()
        d_att = torch.empty((ctx.gat.num_v, 2), dtype=torch.float32, device=dy.device)
        d_x = torch.empty_like(x)
        ctx.gat.run_bwd(y, dy, newval, div, x, d_att, d_x, ctx.slope)
        return d_x, d_att, None, None, None


def gcn_aggregate(agg, x, mode="balanced"):
    """Differentiable y = A.x with an Aggregator_GCN (square graph, sum reduction)."""
    return _GcnAggregate.apply(x, agg, mode)


def gat_aggregate(gat, x, att, mode="balanced", slope=0.2):
    """Differentiable single-head fused GAT aggregation with an Aggregator_GAT; att is [V,2] (centre, source terms).
    `mode` must be one whose edge order is the CSR order (rows or balanced on the chunked plan): newval is indexed by edge."""
    if mode == "balanced" and gat.balanced_partitions() > 0:
        mode = 0  # the source-partitioned order permutes the edges; the canonical rows mode keeps newval in CSR order
    return _GatAggregate.apply(x, att, gat, mode, slope)

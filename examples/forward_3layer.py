#!/usr/bin/env python3
"""3-layer GCN / GAT forward (512 -> 128 -> 64 -> 32) on top of the aggregation library -- the harness of the
reference's Figure7/our.py (layers :171-188, timing loop :247-263: 100 warm-up + 100 timed iterations), written
against the same extension function names (gcn_init / gcn_schedule / gcn_run / gat_init / gat_run ...).

    python examples/forward_3layer.py --model our_GCN --dataset arxiv [--datadir DIR --reorder _thres_0.2]

Without --datadir a synthetic arxiv/reddit/products-shaped CSR is generated (the reference's data sets are an
external download).  The dense layers are the library's own f32-MFMA GEMM by default (--dense torch: torch.mm = rocBLAS /
hipBLASLt, exactly as in the reference script)."""
import argparse
import json
import sys
import os
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc  # noqa: E402


DIMS = [512, 128, 64, 32]                                    # our.py:92-95


class Model:
    """The aggregators, weights and layer functions of Figure7/our.py for one graph (ptrs, idxs: int32 device CSR)."""

    def __init__(self, ptrs, idxs, neighbor_num=32, sched=1, fused_relu=False, dense=torch.mm, seed=123):
        dev = ptrs.device
        torch.manual_seed(seed)                               # our.py:76
        self.num_v, self.num_e = ptrs.numel() - 1, idxs.numel()
        self.vals = torch.ones(self.num_e, device=dev)        # our.py:78
        self.at = gnc.gcn_init(ptrs, idxs, self.vals)
        gnc.gcn_schedule(self.at, neighbor_num)               # our.py:84
        self.at_gat = gnc.gat_init(ptrs, idxs)
        gnc.gat_schedule(self.at_gat, neighbor_num)
        self.sched, self.fused_relu, self.dense = sched, fused_relu, dense
        # 1/sqrt(fan_in) scaling keeps activations O(1): the reference's GAT kernel exponentiates raw scores without a
        # max-subtraction (aggr_gat.h:138-143), so un-scaled randn weights overflow exp() in the deeper layers
        self.weights = [torch.randn(DIMS[k], DIMS[k + 1], device=dev) / DIMS[k] ** 0.5 for k in range(3)]
        self.weights_lr = [torch.randn(DIMS[k + 1], 2, device=dev) / DIMS[k + 1] ** 0.5 for k in range(3)]
        self.h = torch.randn(self.num_v, DIMS[0], device=dev)
        self.outs = [torch.empty(self.num_v, DIMS[k + 1], device=dev) for k in range(3)]
        self.trace = None                                     # set to a list to record every layer's intermediates

    def gcn_layer(self, feat, out, w):                        # our.py:171-176
        feat2 = self.dense(feat, w)
        if self.fused_relu:
            gnc.gcn_run(self.at, feat2, out, 128, self.sched, relu=True)
            res = out
        else:
            gnc.gcn_run(self.at, feat2, out, 128, self.sched)
            res = F.relu(out)
        if self.trace is not None:
            self.trace.append(dict(feat=feat, w=w, feat2=feat2, out=res.clone()))
        return res

    def gat_layer(self, feat, out, w, w_lr):                  # our.py:179-188
        feat2 = self.dense(feat, w)
        att_lr = self.dense(feat2, w_lr)
        gnc.gat_run(self.at_gat, feat2, att_lr, out, 128, self.sched)
        if self.trace is not None:
            self.trace.append(dict(feat=feat, w=w, w_lr=w_lr, feat2=feat2, att=att_lr, out=out.clone()))
        return out

    def forward(self, model="our_GCN"):
        x = self.h
        for k in range(3):
            x = (self.gcn_layer(x, self.outs[k], self.weights[k]) if model == "our_GCN"
                 else self.gat_layer(x, self.outs[k], self.weights[k], self.weights_lr[k]))
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="our_GCN", choices=["our_GCN", "our_GAT"])
    ap.add_argument("--dataset", default="arxiv")
    ap.add_argument("--datadir", default=None)
    ap.add_argument("--reorder", default="")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--neighbor-num", type=int, default=32)   # our.py:84
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--balanced", action="store_true",
                    help="use the library-balanced mode instead of the reference's scheduled=1 on its neighbor-grouping "
                         "schedule (on high-degree graphs that is the source-partitioned order)")
    ap.add_argument("--fused-relu", action="store_true",
                    help="GCN: apply the ReLU inside the aggregation kernel (GNNAGG_FLAG_RELU) instead of F.relu")
    ap.add_argument("--dense", default="library", choices=["library", "torch"],
                    help="dense layers: the library's f32-MFMA GEMM (gnnagg_matmul_nn: bit-exact against the oracle, every stage of "
                         "the forward then is; 512 -> 128: 261 us vs rocBLAS 239 us) or torch.mm as the reference script uses")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture one forward in a HIP graph and replay it (the 9-12 launches of a forward are short "
                         "enough on the arxiv-sized graph for launch gaps to show)")
    args = ap.parse_args()
    dev = torch.device("cuda", args.gpu)

    if args.datadir:
        ptrs, idxs = gnc.new_load(args.dataset, args.reorder, args.gpu, datadir=args.datadir)
    else:
        ptrs, idxs = gnc.graph.dataset(args.dataset, device=dev)
    m = Model(ptrs, idxs, args.neighbor_num, "balanced" if args.balanced else 1, args.fused_relu,
              gnc.matmul_NN if args.dense == "library" else torch.mm)
    num_v, num_e = m.num_v, m.num_e

    def forward():
        return m.forward(args.model)

    step, result = forward, None
    if args.hip_graph:
        forward()                                            # plans, scratch and rocBLAS workspaces exist before capture
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            forward()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            result = forward()
        step = graph.replay
    for _ in range(args.iters):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    y = forward()
    if result is not None:
        assert torch.equal(result, y), "graph replay differs from the eager forward"
    print(json.dumps({"model": args.model, "dataset": args.dataset, "num_v": num_v, "num_e": num_e,
                      "seconds_per_forward": dt, "hip_graph": bool(args.hip_graph), "fused_relu": bool(args.fused_relu), "balanced": bool(args.balanced), "dense": args.dense, "finite": bool(torch.isfinite(y).all().item())}))


if __name__ == "__main__":
    main()

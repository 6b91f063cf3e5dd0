#!/usr/bin/env python3
"""All-to-all-v micro-benchmark of the halo exchange pattern over RCCL / xGMI, both transports:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/rccl_alltoallv_bench.py [MB ...]
Every rank sends `MB` megabytes to each of the other ranks (the weak-scaling step at N = 8 moves ~21 MB per pair).  Prints,
per size, the time of torch.distributed.all_to_all_single and of the C-ABI's grouped ncclSend / ncclRecv
(gnnagg_dist_alltoallv), and the per-link-direction rate = bytes sent to ONE peer / time."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_computing_amd.dist import RcclTransport  # noqa: E402

sizes_mb = [float(a) for a in sys.argv[1:]] or [1, 4, 16, 21, 64]
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
rccl = RcclTransport(None, dev)


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    t = torch.tensor([a.elapsed_time(b) * 1e-3 / iters], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


for mb in sizes_mb:
    n = int(mb * 1e6 / 4)
    send = torch.randn(n * world, device=dev)
    recv = torch.empty_like(send)
    counts = [n] * world
    t_torch = timed(lambda: dist.all_to_all_single(recv, send, counts, counts))
    t_cabi = timed(lambda: rccl.alltoallv(send, counts, recv, counts))
    if rank == 0:
        print(json.dumps({"world": world, "MB_per_pair": mb, "torch_all_to_all_us": t_torch * 1e6, "cabi_grouped_sendrecv_us": t_cabi * 1e6,
                          "per_link_GBps_torch": mb / 1e3 / t_torch if world > 1 else None,
                          "per_link_GBps_cabi": mb / 1e3 / t_cabi if world > 1 else None}), flush=True)
dist.barrier()
rccl.close()
dist.destroy_process_group()

"""1-D row-partitioned aggregation across the GPUs of one node (one process per GPU).

No reference counterpart: the reference asserts a single GPU (Figure9/main.cu:19).  Design
(SURVEY.md 8e): rank g owns a contiguous, nnz-balanced block of CSR rows and the X/Y rows of the
same node range.  Columns outside the block are "halo" nodes; their feature rows are pulled from
their owners once per aggregation with ONE all-to-all-v (RCCL over xGMI -- each pairwise message
rides its own direct link) into the tail of an extended feature buffer
X_ext = [X_local ; X_halo], and the local CSR's column ids are pre-translated into X_ext slots so the
aggregation kernel is exactly the single-GPU one.  Per-row accumulation order is the global CSR order,
so results are bit-identical to the single-GPU run.

torch.distributed is the transport only (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests);
the send-buffer pack and the aggregation are HIP kernels behind the C-ABI.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .aggregator import Aggregator_GAT, Aggregator_GCN, halo_plan, partition_rows
from ._lib import check, lib


def _hip_pack_rows(x, ids, out):
    """out[i,:] = x[ids[i],:] with the HIP kernel (gnnagg_pack_rows)."""
    if not x.is_cuda:
        raise RuntimeError("halo pack runs on the GPU only (no CPU fallback); inject pack_fn in CPU tests")
    check(lib().gnnagg_pack_rows(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ids.data_ptr()), int(ids.numel()),
                                 int(x.shape[1]), ctypes.c_void_p(out.data_ptr()),
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


class HaloExchange:
    """Static communication plan of one rank + the per-aggregation exchange.

    Built from the *global* CSR (every rank holds it at plan time, as every rank of the reference's
    drivers loads the whole graph file); only the local slice is kept afterwards.
    """

    def __init__(self, ptr, idx, rank=None, world=None, group=None, device="cpu", bounds=None, pack_fn=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = torch.device(device)
        self.pack_fn = pack_fn or _hip_pack_rows
        ptr = np.ascontiguousarray(ptr, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.bounds = partition_rows(ptr, self.world) if bounds is None else np.asarray(bounds, np.int32)
        plan = halo_plan(ptr, idx, self.bounds, self.rank)
        self.n_local = plan["n_local"]
        self.local_ptr = plan["local_ptr"]
        self.local_idx = plan["local_idx"]
        self.halo_ids = plan["halo_ids"]            # global ids, grouped by owner, ascending
        self.recv_counts = plan["halo_counts"].astype(np.int64)  # rows received from each rank
        self.n_halo = int(len(self.halo_ids))
        self.row0 = int(self.bounds[self.rank])
        self.e0, self.e1 = int(ptr[self.bounds[self.rank]]), int(ptr[self.bounds[self.rank + 1]])
        # one-time: tell every owner which of its rows this rank needs
        self._exchange_requests()

    def _exchange_requests(self):
        w = self.world
        cpu_like = self.device if self.device.type == "cuda" else torch.device("cpu")
        recv_counts = torch.tensor(self.recv_counts, dtype=torch.int64, device=cpu_like)
        send_counts = torch.empty(w, dtype=torch.int64, device=cpu_like)
        if w > 1:
            dist.all_to_all_single(send_counts, recv_counts, group=self.group)
        else:
            send_counts.copy_(recv_counts)
        self.send_counts = send_counts.cpu().numpy().astype(np.int64)
        req = torch.from_numpy(self.halo_ids.astype(np.int32)).to(cpu_like)
        serve = torch.empty(int(self.send_counts.sum()), dtype=torch.int32, device=cpu_like)
        if w > 1:
            dist.all_to_all_single(serve, req, output_split_sizes=self.send_counts.tolist(),
                                   input_split_sizes=self.recv_counts.tolist(), group=self.group)
        # rows of the LOCAL x to pack, in the order the peers expect them
        self.send_ids = (serve - self.row0).to(self.device)
        self.n_send = int(self.send_ids.numel())
        assert self.n_send == 0 or (int(self.send_ids.min()) >= 0 and int(self.send_ids.max()) < self.n_local)

    def halo_bytes(self, feat):
        return self.n_halo * feat * 4

    def alloc_x_ext(self, feat, dtype=torch.float32):
        return torch.empty((self.n_local + self.n_halo, feat), dtype=dtype, device=self.device)

    def exchange(self, x_ext, send_buf=None):
        """Fills x_ext[n_local:] with the halo rows (x_ext[:n_local] holds this rank's rows)."""
        feat = x_ext.shape[1]
        if self.world == 1:
            return x_ext
        if send_buf is None or send_buf.shape[0] < self.n_send:
            send_buf = torch.empty((max(self.n_send, 1), feat), dtype=x_ext.dtype, device=x_ext.device)
        if self.n_send:
            self.pack_fn(x_ext[:self.n_local], self.send_ids, send_buf)
        dist.all_to_all_single(x_ext[self.n_local:], send_buf[:self.n_send],
                               output_split_sizes=self.recv_counts.tolist(),
                               input_split_sizes=self.send_counts.tolist(), group=self.group)
        return x_ext


class PartitionedGCN:
    """Row-partitioned GCN/SAGE aggregation: y_local = A[rows of this rank, :] @ X (global)."""

    def __init__(self, ptr, idx, val=None, feat=128, group=None, device=None, mode="balanced", rank=None, world=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device)
        hx = self.hx
        self.feat, self.mode = feat, mode
        self.d_ptr = torch.from_numpy(hx.local_ptr).to(device)
        self.d_idx = torch.from_numpy(hx.local_idx).to(device)
        self.d_val = None if val is None else torch.from_numpy(
            np.ascontiguousarray(np.asarray(val, np.float32)[hx.e0:hx.e1])).to(device)
        self.agg = Aggregator_GCN(self.d_ptr, self.d_idx, self.d_val, feat, feat)
        if mode == "balanced":
            self.agg.schedule_balanced(0)
        self.x_ext = hx.alloc_x_ext(feat)
        self.send_buf = torch.empty((max(hx.n_send, 1), feat), dtype=torch.float32, device=device)
        self.y = torch.empty((hx.n_local, feat), dtype=torch.float32, device=device)
        self.num_e_local = hx.e1 - hx.e0

    def set_local_x(self, x_local):
        self.x_ext[:self.hx.n_local].copy_(x_local)

    def step(self, reduce="sum"):
        """One aggregation: halo all-to-all, then the single-GPU kernel on [X_local ; X_halo]."""
        self.hx.exchange(self.x_ext, self.send_buf)
        self.agg.run(self.x_ext, self.y, 512, self.mode, reduce=reduce)
        return self.y


class PartitionedGAT:
    """Row-partitioned fused GAT; att rows travel with the feature rows (one extra exchange)."""

    def __init__(self, ptr, idx, feat=256, heads=8, group=None, device=None, mode="balanced", rank=None, world=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device)
        hx = self.hx
        self.feat, self.heads, self.mode = feat, heads, mode
        self.d_ptr = torch.from_numpy(hx.local_ptr).to(device)
        self.d_idx = torch.from_numpy(hx.local_idx).to(device)
        self.agg = Aggregator_GAT(self.d_ptr, self.d_idx, feat, feat)
        if mode == "balanced":
            self.agg.schedule_balanced(0)
        self.x_ext = hx.alloc_x_ext(feat)
        self.att_ext = hx.alloc_x_ext(heads * 2)
        self.y = torch.empty((hx.n_local, feat), dtype=torch.float32, device=device)

    def step(self, slope=0.2):
        self.hx.exchange(self.x_ext)
        self.hx.exchange(self.att_ext)
        # the kernel indexes att by X_ext slot; only the first n_local rows are destinations
        self.agg.run_with_feat(self.x_ext, self.att_ext, self.y, 128, self.mode, self.feat, self.heads, slope)
        return self.y

"""1-D row-partitioned aggregation across the GPUs of one node (one process per GPU).

No reference counterpart: the reference asserts a single GPU (Figure9/main.cu:19).  Design
(SURVEY.md 8e): rank g owns a contiguous, nnz-balanced block of CSR rows and the X/Y rows of the
same node range.  Columns outside the block are "halo" nodes; their feature rows are pulled from
their owners once per aggregation with ONE all-to-all-v (RCCL over xGMI -- each pairwise message
rides its own direct link).

Two execution plans share the same communication plan:

* ``overlap=False``: the halo lands in the tail of X_ext = [X_local ; X_halo], the local CSR's
  column ids are pre-translated into X_ext slots and the single-GPU kernel runs unchanged.  Per-row
  accumulation order is the global CSR order, so results are bit-identical to the single-GPU run.
* ``overlap=True`` (default for sums): the rank's CSR is split into the edges whose source is owned
  (A_loc) and the edges whose source is a halo row (A_rem).  The all-to-all is started
  asynchronously, ``Y = A_loc . X_local`` runs while it is in flight, then ``Y += A_rem . X_halo``
  (GNNAGG_FLAG_ACCUMULATE).  Deterministic; equals the single-GPU result up to the one extra fp32
  add per row that joins the two parts.

torch.distributed is the transport only (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests);
the send-buffer pack and the aggregation are HIP kernels behind the C-ABI.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

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
    ``offline=True`` builds the plan of (rank, world) without any collective (single-process tests):
    the caller fills the halo rows by hand.
    """

    def __init__(self, ptr, idx, rank=None, world=None, group=None, device="cpu", bounds=None, pack_fn=None,
                 offline=False):
        self.group = group
        self.offline = offline
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
        if offline:
            self.send_counts = np.zeros(self.world, np.int64)
            self.send_ids = torch.zeros(0, dtype=torch.int32, device=self.device)
            self.n_send = 0
        else:
            self._exchange_requests()  # one-time: tell every owner which of its rows this rank needs

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

    def split_local_remote(self):
        """(ptr_loc, idx_loc, ptr_rem, idx_rem, edge_is_local): CSR of the edges with an owned source (column ids
        local rows) and of the edges with a halo source (column ids = halo slot, 0-based); in-row order is kept."""
        is_loc = self.local_idx < self.n_local
        rows = np.repeat(np.arange(self.n_local), np.diff(self.local_ptr))

        def sub(mask, shift):
            cnt = np.bincount(rows[mask], minlength=self.n_local)
            p = np.zeros(self.n_local + 1, np.int32)
            p[1:] = np.cumsum(cnt)
            return p, (self.local_idx[mask] - shift).astype(np.int32)

        pl, il = sub(is_loc, 0)
        pr, ir = sub(~is_loc, self.n_local)
        return pl, il, pr, ir, is_loc

    def exchange(self, x_local, x_halo, send_buf=None, async_op=False):
        """Fills x_halo[n_halo, F] with the halo rows; x_local[n_local, F] holds this rank's rows.
        Returns the work handle when async_op (None for a single rank / offline plan)."""
        if self.offline or (self.world == 1 and not (dist.is_available() and dist.is_initialized())):
            return None
        feat = x_local.shape[1]
        if send_buf is None or send_buf.shape[0] < self.n_send:
            send_buf = torch.empty((max(self.n_send, 1), feat), dtype=x_local.dtype, device=x_local.device)
        if self.n_send:
            self.pack_fn(x_local, self.send_ids, send_buf)
        return dist.all_to_all_single(x_halo, send_buf[:self.n_send],
                                      output_split_sizes=self.recv_counts.tolist(),
                                      input_split_sizes=self.send_counts.tolist(), group=self.group, async_op=async_op)


class PartitionedGCN:
    """Row-partitioned GCN/SAGE aggregation: y_local = A[rows of this rank, :] @ X (global)."""

    def __init__(self, ptr, idx, val=None, feat=128, group=None, device=None, mode="balanced", rank=None, world=None,
                 overlap=True, offline=False):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device, offline=offline)
        hx = self.hx
        self.feat, self.mode = feat, mode
        self.overlap = bool(overlap) and mode == "balanced"
        val_loc = None if val is None else np.ascontiguousarray(np.asarray(val, np.float32)[hx.e0:hx.e1])
        self.x_ext = hx.alloc_x_ext(feat)                # [X_local ; X_halo], one allocation
        self.x_local = self.x_ext[:hx.n_local]
        self.x_halo = self.x_ext[hx.n_local:]
        self.send_buf = torch.empty((max(hx.n_send, 1), feat), dtype=torch.float32, device=device)
        self.y = torch.empty((hx.n_local, feat), dtype=torch.float32, device=device)
        self.num_e_local = hx.e1 - hx.e0
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        if self.overlap:
            pl, il, pr, ir, is_loc = hx.split_local_remote()
            self.agg_loc = Aggregator_GCN(t(pl), t(il), None if val_loc is None else t(val_loc[is_loc]), feat, feat)
            self.agg_rem = Aggregator_GCN(t(pr), t(ir), None if val_loc is None else t(val_loc[~is_loc]), feat, feat)
            self.agg_loc.schedule_balanced(0)
            self.agg_rem.schedule_balanced(0)
            self.num_e_remote = int(len(ir))
        else:
            self.agg = Aggregator_GCN(t(hx.local_ptr), t(hx.local_idx), None if val_loc is None else t(val_loc), feat, feat)
            if mode == "balanced":
                self.agg.schedule_balanced(0)

    def set_local_x(self, x_local):
        self.x_local.copy_(x_local)

    def compute(self, reduce="sum", work=None):
        """The aggregation kernels of one step, given that the exchange `work` (or None) fills x_halo."""
        if self.overlap and reduce == "sum":
            self.agg_loc.run(self.x_local, self.y, 512, "balanced")          # overlaps the all-to-all
            if work is not None:
                work.wait()                                                    # current stream waits for the halo
            if self.hx.n_halo:
                self.agg_rem.run(self.x_halo, self.y, 512, "balanced", accumulate=True)
        else:
            if work is not None:
                work.wait()
            if self.overlap:  # mean / max need the whole row at once: fall back to the one-pass plan lazily
                raise NotImplementedError("overlap plan supports reduce='sum'; build with overlap=False for mean/max")
            self.agg.run(self.x_ext, self.y, 512, self.mode, reduce=reduce)
        return self.y

    def step(self, reduce="sum"):
        """One aggregation: halo all-to-all (asynchronous when overlapping) + kernels."""
        work = self.hx.exchange(self.x_local, self.x_halo, self.send_buf, async_op=self.overlap)
        return self.compute(reduce, work)


class PartitionedGAT:
    """Row-partitioned fused GAT; att rows travel with the feature rows (one extra exchange)."""

    def __init__(self, ptr, idx, feat=256, heads=8, group=None, device=None, mode="balanced", rank=None, world=None,
                 offline=False):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device, offline=offline)
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

    def compute(self, slope=0.2):
        # the kernel indexes att by X_ext slot; only the first n_local rows are destinations
        self.agg.run_with_feat(self.x_ext, self.att_ext, self.y, 128, self.mode, self.feat, self.heads, slope)
        return self.y

    def step(self, slope=0.2):
        n = self.hx.n_local
        self.hx.exchange(self.x_ext[:n], self.x_ext[n:])
        self.hx.exchange(self.att_ext[:n], self.att_ext[n:])
        return self.compute(slope)

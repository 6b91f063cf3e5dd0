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

Transports (same plan, same buffers): "torch" = torch.distributed.all_to_all_single (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests); "rccl" = the C-ABI's own grouped ncclSend / ncclRecv (gnnagg_dist_halo_exchange)
-- what a C++ driver uses (drivers/dist_step.cpp); torch.distributed then only carries the 128-byte unique id.  The
send-buffer pack and the aggregation are HIP kernels behind the C-ABI either way.
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


class _StreamWork:
    """What exchange(async_op=True) returns for the C-ABI transport: wait() makes the current stream wait for the
    communication stream (the counterpart of torch.distributed's Work.wait())."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


class RcclTransport:
    """The C-ABI's RCCL communicator (gnnagg_dist_*): created from a unique id that rank 0 draws and torch.distributed
    broadcasts (any launcher channel would do: drivers/dist_step.cpp passes it through a file)."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(lib().gnnagg_dist_unique_id(buf))
        holder = [bytes(buf.raw)]
        dist.broadcast_object_list(holder, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = ctypes.c_int64(0)
        with torch.cuda.device(dev):
            check(lib().gnnagg_dist_comm_create(ctypes.c_char_p(holder[0]), self.rank, self.world, ctypes.byref(self._h)))
        self.stream = torch.cuda.Stream(device=dev)

    def close(self):
        if self._h.value:
            lib().gnnagg_dist_comm_destroy(self._h)
            self._h = ctypes.c_int64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alltoallv(self, send, send_counts, recv, recv_counts):
        """Contiguous device tensors, counts in elements of send.element_size() per rank; on the current stream."""
        n = self.world
        sc = (ctypes.c_longlong * n)(*[int(v) for v in send_counts])
        rc = (ctypes.c_longlong * n)(*[int(v) for v in recv_counts])
        check(lib().gnnagg_dist_alltoallv(self._h, ctypes.c_void_p(send.data_ptr()), sc, ctypes.c_void_p(recv.data_ptr()), rc,
                                          int(send.element_size()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def halo_exchange(self, x_local, send_ids, send_rows, recv_rows, send_buf, x_halo, async_op):
        n = self.world
        sc = (ctypes.c_longlong * n)(*[int(v) for v in send_rows])
        rc = (ctypes.c_longlong * n)(*[int(v) for v in recv_rows])
        st = self.stream if async_op else torch.cuda.current_stream()
        if async_op:
            st.wait_stream(torch.cuda.current_stream())      # x_local is ready when the pack starts
        check(lib().gnnagg_dist_halo_exchange(self._h, ctypes.c_void_p(x_local.data_ptr()), ctypes.c_void_p(send_ids.data_ptr()),
                                              sc, rc, int(x_local.shape[1]), ctypes.c_void_p(send_buf.data_ptr()),
                                              ctypes.c_void_p(x_halo.data_ptr()), ctypes.c_void_p(st.cuda_stream)))
        return _StreamWork(st) if async_op else None


class HaloExchange:
    """Static communication plan of one rank + the per-aggregation exchange.

    Built from the *global* CSR (every rank holds it at plan time, as every rank of the reference's
    drivers loads the whole graph file); only the local slice is kept afterwards.
    ``offline=True`` builds the plan of (rank, world) without any collective (single-process tests):
    the caller fills the halo rows by hand.
    """

    def __init__(self, ptr, idx, rank=None, world=None, group=None, device="cpu", bounds=None, pack_fn=None,
                 offline=False, row_slice=False, num_cols=None, transport="torch"):
        """row_slice=False: (ptr, idx) is the global CSR (every rank holds it at plan time).  row_slice=True: (ptr, idx) are
        THIS rank's rows only -- ptr[0 .. n_local] with any base offset, idx with global column ids -- and `bounds`
        (partition_rows of the global ptr, e.g. computed by rank 0 and broadcast) and `num_cols` are required.
        transport: "torch" (torch.distributed all_to_all_single) or "rccl" (the C-ABI's grouped send/recv)."""
        self.group = group
        self.offline = offline
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = torch.device(device)
        self.pack_fn = pack_fn or _hip_pack_rows
        self.rccl = None
        if transport == "rccl" and not offline and self.world > 1:
            self.rccl = RcclTransport(group, self.device)
        elif transport not in ("torch", "rccl"):
            raise ValueError("transport must be 'torch' or 'rccl'")
        ptr = np.ascontiguousarray(ptr, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if row_slice:
            if bounds is None or num_cols is None:
                raise ValueError("row_slice=True needs bounds and num_cols")
            self.bounds = np.asarray(bounds, np.int32)
            plan = halo_plan(ptr, idx, self.bounds, self.rank, row_slice=True, num_cols=int(num_cols))
            self.e0 = int(ptr[0])
            self.e1 = int(ptr[-1])
        else:
            self.bounds = partition_rows(ptr, self.world) if bounds is None else np.asarray(bounds, np.int32)
            plan = halo_plan(ptr, idx, self.bounds, self.rank)
            self.e0, self.e1 = int(ptr[self.bounds[self.rank]]), int(ptr[self.bounds[self.rank + 1]])
        self.n_local = plan["n_local"]
        self.local_ptr = plan["local_ptr"]
        self.local_idx = plan["local_idx"]
        self.halo_ids = plan["halo_ids"]            # global ids, grouped by owner, ascending
        self.recv_counts = plan["halo_counts"].astype(np.int64)  # rows received from each rank
        self.n_halo = int(len(self.halo_ids))
        self.row0 = int(self.bounds[self.rank])
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
        if self.rccl is not None:
            return self.rccl.halo_exchange(x_local, self.send_ids, self.send_counts, self.recv_counts, send_buf, x_halo, async_op)
        if self.n_send:
            self.pack_fn(x_local, self.send_ids, send_buf)
        return dist.all_to_all_single(x_halo, send_buf[:self.n_send],
                                      output_split_sizes=self.recv_counts.tolist(),
                                      input_split_sizes=self.send_counts.tolist(), group=self.group, async_op=async_op)


class PartitionedGCN:
    """Row-partitioned GCN/SAGE aggregation: y_local = A[rows of this rank, :] @ X (global)."""

    def __init__(self, ptr, idx, val=None, feat=128, group=None, device=None, mode="balanced", rank=None, world=None,
                 overlap=True, offline=False, row_slice=False, bounds=None, num_cols=None, transport="torch"):
        """row_slice=True: ptr / idx / val hold this rank's rows only (see HaloExchange)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device, offline=offline,
                               row_slice=row_slice, bounds=bounds, num_cols=num_cols, transport=transport)
        hx = self.hx
        self.feat, self.mode = feat, mode
        self.overlap = bool(overlap) and mode == "balanced"
        if val is None:
            val_loc = None
        elif row_slice:
            val_loc = np.ascontiguousarray(np.asarray(val, np.float32))
        else:
            val_loc = np.ascontiguousarray(np.asarray(val, np.float32)[hx.e0:hx.e1])
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

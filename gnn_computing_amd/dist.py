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

Staged exchange (round 4; ``stages=``): the halo rows arrive in S stages, stage s with its own all-to-all-v, and the halo-source
edges are split by the stage their source arrives in -- the pass over stage s's edges runs while stage s + 1 is on the links.
Buffers are stage-major (send buffer, halo tail: stage 0's rows in rank order, then stage 1's ...), so every stage is a plain
all-to-all-v on a contiguous sub-range.  ``("stripe", K)``: every stage takes 1/K of EVERY peer's rows -- on the point-to-point
xGMI mesh each peer pair has its own link, so a stage must talk to all peers to keep all links busy (the default for
``stages="auto"``: K from the halo bytes); ``"owner"``: stage s exchanges with the peers at ring distance s + 1 only (one link per
stage: for switched fabrics and for tests).  Result: y = local pass, += stage 0's pass, += stage 1's ... in that fixed order.

Transports (same plan, same buffers): "torch" = torch.distributed.all_to_all_single (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests); "rccl" = the C-ABI's own grouped ncclSend / ncclRecv (gnnagg_dist_halo_exchange)
-- what a C++ driver uses (drivers/dist_step.cpp); torch.distributed then only carries the 128-byte unique id.  The
send-buffer pack and the aggregation are HIP kernels behind the C-ABI either way.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from .aggregator import REDUCE, Aggregator_GAT, Aggregator_GCN, halo_plan, partition_rows
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


class _EventWork:
    """one stage of a staged exchange on the C-ABI transport: wait() orders the current stream behind the stage's event"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


class _StagedWork:
    """the stages' work handles in order; wait() waits for all of them, stage(s).wait() for one"""

    def __init__(self, works):
        self.works = works

    def stage(self, s):
        return self.works[s]

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()


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

    def transport_info(self):
        """(library the nccl* entry points were loaded from, True if that was the GNNAGG_RCCL_LIB override, PCI bus id of the device this
        communicator is bound to) -- gnnagg_dist_transport_info; what a measurement over this communicator must state."""
        path, bus, over = ctypes.create_string_buffer(1024), ctypes.create_string_buffer(64), ctypes.c_int(0)
        check(lib().gnnagg_dist_transport_info(self._h, path, 1024, bus, 64, ctypes.byref(over)))
        return path.value.decode(), bool(over.value), bus.value.decode()

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
                 offline=False, row_slice=False, num_cols=None, transport="torch", stages=1, row_bytes=512):
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
        self.row_bytes = int(row_bytes)
        self._plan_stages(stages)
        if offline:
            self.send_counts = np.zeros(self.world, np.int64)
            self.stage_send = np.zeros((self.n_stages, self.world), np.int64)
            self.stage_send0 = np.zeros(self.n_stages + 1, np.int64)
            self.send_ids = torch.zeros(0, dtype=torch.int32, device=self.device)
            self.n_send = 0
        else:
            self._exchange_requests()  # one-time: tell every owner which of its rows this rank needs

    def _stage_plan(self, recv_rows=None, send_rows=None):
        """gnnagg_halo_stage_plan (host C++, host_graph.cpp): the stage counts and slot / send-order permutations of this rank"""
        w = self.world
        mode = 1 if self.stage_mode == "owner" else 0
        ns = ctypes.c_int(0)
        check(lib().gnnagg_halo_stage_plan(None, None, w, self.rank, mode, max(self.stage_k, 1), ctypes.byref(ns), None, None, None, None))
        S = ns.value
        out = {"n_stages": S}
        ll = lambda a: np.ascontiguousarray(a, dtype=np.int64)   # noqa: E731
        if recv_rows is not None:
            rr = ll(recv_rows)
            st, perm = np.zeros((S, w), np.int64), np.zeros(max(int(rr.sum()), 1), np.int32)
            check(lib().gnnagg_halo_stage_plan(rr.ctypes.data, None, w, self.rank, mode, max(self.stage_k, 1), ctypes.byref(ns), st.ctypes.data,
                                               perm.ctypes.data, None, None))
            out["stage_recv"], out["new_of_old"] = st, perm[:int(rr.sum())]
        if send_rows is not None:
            sr = ll(send_rows)
            st, order = np.zeros((S, w), np.int64), np.zeros(max(int(sr.sum()), 1), np.int32)
            check(lib().gnnagg_halo_stage_plan(None, sr.ctypes.data, w, self.rank, mode, max(self.stage_k, 1), ctypes.byref(ns), None, None,
                                               st.ctypes.data, order.ctypes.data))
            out["stage_send"], out["send_order"] = st, order[:int(sr.sum())]
        return out

    def _plan_stages(self, stages):
        """Chooses the stages and renumbers the halo slots stage-major: halo_ids, the halo columns of local_idx and (later) the
        send list follow.  stages: 1 | K | ("stripe", K) | "owner" | "auto" (stripes, one per 64 MB of halo rows -- `row_bytes` per row --
        at most 4: a stage is one more collective launch and one more pass over the rows of y, worth it when the transfer it hides
        under is long; ("auto", bytes) sets another size) -- from the largest halo of the job, so that all ranks agree."""
        w = self.world
        if stages == "auto" or (isinstance(stages, tuple) and stages[0] == "auto"):
            # every rank must cut the lists the same way: the stage count follows the LARGEST halo of the job, not this rank's
            per_stage = (64 << 20) if stages == "auto" else int(stages[1])
            n = self.n_halo
            if not self.offline and w > 1:
                t = torch.tensor([n], dtype=torch.int64, device=self.device if self.device.type == "cuda" and dist.get_backend(self.group) == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                n = int(t.item())
            stages = ("stripe", int(min(4, max(1, (n * self.row_bytes) // per_stage))))
        if isinstance(stages, int):
            stages = ("stripe", stages)
        if stages == "owner":
            stages = ("owner", 0)
        mode, k = stages
        if mode not in ("stripe", "owner") or (mode == "stripe" and k < 1):
            raise ValueError("stages: 1, K, ('stripe', K), 'owner' or 'auto'")
        if w == 1 or (mode == "stripe" and k == 1):
            mode, k = "stripe", 1
        self.stage_mode, self.stage_k = mode, k
        plan = self._stage_plan(recv_rows=self.recv_counts)
        self.stage_recv, self.n_stages = plan["stage_recv"], plan["n_stages"]        # [S][owner]
        self.stage_recv0 = np.concatenate([[0], np.cumsum(self.stage_recv.sum(axis=1))]).astype(np.int64)   # first halo slot of a stage
        self.stage_of_slot = np.repeat(np.arange(self.n_stages, dtype=np.int32), self.stage_recv.sum(axis=1))
        if self.n_stages == 1:
            return
        # old slot order: owner-major, ascending id.  new: stage-major, then owner, then ascending id
        new_of_old = plan["new_of_old"].astype(np.int64)
        ids = np.empty_like(self.halo_ids)
        ids[new_of_old] = self.halo_ids
        self.halo_ids = ids
        is_halo = self.local_idx >= self.n_local
        li = self.local_idx.copy()
        li[is_halo] = (new_of_old[self.local_idx[is_halo] - self.n_local] + self.n_local).astype(li.dtype)
        self.local_idx = li
        # the request lists go out owner-major (one all-to-all), each owner's list in the order its rows will arrive (stage by stage):
        # the owner-major slots sorted by their new slot inside every owner = argsort of new_of_old per owner = its inverse restricted
        owner_of_old = np.repeat(np.arange(w), self.recv_counts)
        self._req_order = np.lexsort((new_of_old, owner_of_old))      # positions in the OLD (owner-major) list ...
        self._req_order = new_of_old[self._req_order]                 # ... as indices into the renumbered halo_ids

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
        req_ids = self.halo_ids if self.n_stages == 1 else self.halo_ids[self._req_order]   # owner-major, arrival order inside
        req = torch.from_numpy(req_ids.astype(np.int32)).to(cpu_like)
        serve = torch.empty(int(self.send_counts.sum()), dtype=torch.int32, device=cpu_like)
        if w > 1:
            dist.all_to_all_single(serve, req, output_split_sizes=self.send_counts.tolist(),
                                   input_split_sizes=self.recv_counts.tolist(), group=self.group)
        # rows of the LOCAL x to pack, in the order the peers expect them: stage-major, reader by reader inside a stage.  What
        # reader q receives from this rank in stage s is what stage_counts says for q's list of this rank's rows
        plan = self._stage_plan(send_rows=self.send_counts)
        self.stage_send = plan["stage_send"]
        if self.n_stages > 1:
            serve = serve[torch.from_numpy(plan["send_order"].astype(np.int64)).to(serve.device)]
        self.stage_send0 = np.concatenate([[0], np.cumsum(self.stage_send.sum(axis=1))]).astype(np.int64)
        self.send_ids = (serve - self.row0).to(self.device)
        self.n_send = int(self.send_ids.numel())
        self._split_lists = None
        assert self.n_send == 0 or (int(self.send_ids.min()) >= 0 and int(self.send_ids.max()) < self.n_local)

    def split_lists(self, stage=None):
        """(rows received from, rows sent to) every rank as Python lists -- what all_to_all_single wants; made once, not per step.
        stage=None: the whole exchange (one stage); else that stage's share."""
        if getattr(self, "_split_lists", None) is None:
            self._split_lists = {None: ([int(v) for v in self.recv_counts], [int(v) for v in self.send_counts])}
            for st in range(self.n_stages):
                self._split_lists[st] = ([int(v) for v in self.stage_recv[st]], [int(v) for v in self.stage_send[st]])
        return self._split_lists[stage]

    def split_remote_stages(self):
        """The halo-source edges per stage: [(ptr_s, idx_s, mask_s)] -- CSR over this rank's rows of the edges whose source arrives
        in stage s (column ids = halo slots, 0-based, over the WHOLE halo tail), mask_s over the rank's edge list; in-row order kept."""
        rows = np.repeat(np.arange(self.n_local), np.diff(self.local_ptr))
        is_rem = self.local_idx >= self.n_local
        slot = np.where(is_rem, self.local_idx - self.n_local, 0)
        st_of_edge = np.where(is_rem, self.stage_of_slot[slot] if self.n_halo else 0, -1)
        out = []
        for st in range(self.n_stages):
            m = st_of_edge == st
            p = np.zeros(self.n_local + 1, np.int32)
            p[1:] = np.cumsum(np.bincount(rows[m], minlength=self.n_local))
            out.append((p, slot[m].astype(np.int32), m))
        return out

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
        if self.rccl is not None and self.n_stages == 1:
            return self.rccl.halo_exchange(x_local, self.send_ids, self.send_counts, self.recv_counts, send_buf, x_halo, async_op)
        if self.n_send:
            self.pack_fn(x_local, self.send_ids, send_buf)
        if self.n_stages == 1:
            return dist.all_to_all_single(x_halo, send_buf[:self.n_send],
                                          output_split_sizes=self.split_lists()[0],
                                          input_split_sizes=self.split_lists()[1], group=self.group, async_op=async_op)
        works = [self.exchange_stage(st, send_buf, x_halo, async_op) for st in range(self.n_stages)]
        return _StagedWork(works) if async_op else None

    def exchange_stage(self, st, send_buf, recv_rows, async_op):
        """Stage st of a staged exchange: rows [stage_send0[st], stage_send0[st + 1]) of the (packed, stage-major) send buffer to the
        peers, rows [stage_recv0[st], stage_recv0[st + 1]) of `recv_rows` from them."""
        s0, s1 = int(self.stage_send0[st]), int(self.stage_send0[st + 1])
        r0, r1 = int(self.stage_recv0[st]), int(self.stage_recv0[st + 1])
        if self.rccl is not None:
            stream = self.rccl.stream if async_op else torch.cuda.current_stream()
            if async_op and st == 0:
                stream.wait_stream(torch.cuda.current_stream())
            w = send_buf.shape[1]
            with torch.cuda.stream(stream):
                self.rccl.alltoallv(send_buf[s0:s1], [int(v) * w for v in self.stage_send[st]], recv_rows[r0:r1],
                                    [int(v) * w for v in self.stage_recv[st]])
                ev = torch.cuda.Event()
                ev.record(stream)
            return _EventWork(ev) if async_op else None
        return dist.all_to_all_single(recv_rows[r0:r1], send_buf[s0:s1], output_split_sizes=self.split_lists(st)[0],
                                      input_split_sizes=self.split_lists(st)[1], group=self.group, async_op=async_op)


class PartitionedGCN:
    """Row-partitioned GCN/SAGE aggregation: y_local = A[rows of this rank, :] @ X (global).

    overlap=True (balanced mode): the local-source edges run while the exchange is in flight, the halo-source edges add to them
    afterwards -- for sums directly (GNNAGG_FLAG_ACCUMULATE), for means with the row's TOTAL degree as the divisor of both passes,
    for maxima as max(old, new) where the first pass folded any edge (gnnagg_set_row_aux).  With the "rccl" transport, or on a
    single rank, the whole step is ONE C-ABI call (gnnagg_dist_step_gcn: pack, grouped send / recv on a communication stream,
    local pass, event wait, halo pass)."""

    def __init__(self, ptr, idx, val=None, feat=128, group=None, device=None, mode="balanced", rank=None, world=None,
                 overlap=True, offline=False, row_slice=False, bounds=None, num_cols=None, transport="torch", stages=1):
        """row_slice=True: ptr / idx / val hold this rank's rows only (see HaloExchange).  stages: the staged exchange of the
        module docstring (overlap plan only): 1, K, ("stripe", K), "owner", "auto"."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if not (bool(overlap) and mode == "balanced"):
            stages = 1
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device, offline=offline,
                               row_slice=row_slice, bounds=bounds, num_cols=num_cols, transport=transport, stages=stages, row_bytes=4 * feat)
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
        self._step = ctypes.c_int64(0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        if self.overlap:
            pl, il, pr, ir, is_loc = hx.split_local_remote()
            self.agg_loc = Aggregator_GCN(t(pl), t(il), None if val_loc is None else t(val_loc[is_loc]), feat, feat)
            self.agg_loc.schedule_balanced(0)
            self.num_e_remote = int(len(ir))
            self.deg_total = t(np.diff(hx.local_ptr).astype(np.int32))   # the row's degree in the whole graph (mean divisor)
            # one aggregator per stage over the halo-source edges whose source arrives in that stage (None: no such edges);
            # deg_before[s] = edges of a row folded before stage s's pass (the guarded max join, gnnagg_set_row_aux)
            self.agg_rem_stages, self.deg_before = [], []
            folded = np.diff(pl).astype(np.int64)
            for ps_, is_, m_ in hx.split_remote_stages():
                self.deg_before.append(t(folded.astype(np.int32)))
                folded = folded + np.diff(ps_)
                if len(is_) == 0:
                    self.agg_rem_stages.append(None)
                    continue
                a = Aggregator_GCN(t(ps_), t(is_), None if val_loc is None else t(val_loc[m_]), feat, feat)
                a.schedule_balanced(0)
                self.agg_rem_stages.append(a)
            self.agg_rem = self.agg_rem_stages[0] if hx.n_stages == 1 else None   # (the one-stage name, kept for callers)
            self.deg_local = self.deg_before[0]
            if device.type == "cuda" and not offline and (hx.rccl is not None or hx.world == 1):
                n, S = hx.world, hx.n_stages
                sc = (ctypes.c_longlong * (n * S))(*[int(v) for v in hx.stage_send.reshape(-1)])
                rc = (ctypes.c_longlong * (n * S))(*[int(v) for v in hx.stage_recv.reshape(-1)])
                hs = (ctypes.c_int64 * S)(*[(a._h.value if a is not None else 0) for a in self.agg_rem_stages])
                check(lib().gnnagg_dist_step_create_staged(hx.rccl._h if hx.rccl is not None else ctypes.c_int64(0), self.agg_loc._h, S,
                                                           hs, ctypes.c_void_p(hx.send_ids.data_ptr()) if hx.n_send else None, sc, rc,
                                                           ctypes.byref(self._step)))
        else:
            self.agg = Aggregator_GCN(t(hx.local_ptr), t(hx.local_idx), None if val_loc is None else t(val_loc), feat, feat)
            if mode == "balanced":
                self.agg.schedule_balanced(0)

    def close(self):
        if self._step.value:
            lib().gnnagg_dist_step_destroy(self._step)
            self._step = ctypes.c_int64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_local_x(self, x_local):
        self.x_local.copy_(x_local)

    def _set_aux(self, reduce):
        """what the two passes need to know about each other's rows (gnnagg_set_row_aux)"""
        kind = reduce if reduce in ("mean", "max") else "sum"
        if getattr(self, "_aux_kind", None) == kind:
            return            # (two host calls saved per step: the step is a handful of launches)
        self._aux_kind = kind
        self.agg_loc.set_row_aux(self.deg_total if reduce == "mean" else None)
        for a, before in zip(self.agg_rem_stages, self.deg_before):
            if a is not None:
                a.set_row_aux(self.deg_total if reduce == "mean" else before if reduce == "max" else None)

    def compute(self, reduce="sum", work=None):
        """The aggregation kernels of one step, given that the exchange `work` (or None) fills x_halo."""
        if self.overlap:
            self._set_aux(reduce)
            self.agg_loc.run(self.x_local, self.y, 512, "balanced", reduce=reduce)   # overlaps the all-to-all
            for st, a in enumerate(self.agg_rem_stages):
                if work is not None:                                                   # current stream waits for stage st of the halo
                    (work.stage(st) if isinstance(work, _StagedWork) else work).wait()
                    if not isinstance(work, _StagedWork):
                        work = None
                if a is not None:
                    a.run(self.x_halo, self.y, 512, "balanced", reduce=reduce, accumulate=True)
        else:
            if work is not None:
                work.wait()
            self.agg.run(self.x_ext, self.y, 512, self.mode, reduce=reduce)
        return self.y

    def step(self, reduce="sum"):
        """One aggregation: halo all-to-all (asynchronous when overlapping) + kernels."""
        if self._step.value:   # ONE host call: pack, grouped send / recv on the step's stream, local pass, wait, halo pass
            self._set_aux(reduce)
            hx = self.hx
            check(lib().gnnagg_dist_step_gcn(self._step, ctypes.c_void_p(self.x_local.data_ptr()),
                                             ctypes.c_void_p(self.x_halo.data_ptr()) if hx.n_halo else None,
                                             ctypes.c_void_p(self.send_buf.data_ptr()), ctypes.c_void_p(self.y.data_ptr()), int(self.feat),
                                             REDUCE[reduce], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return self.y
        work = self.hx.exchange(self.x_local, self.x_halo, self.send_buf, async_op=self.overlap)
        return self.compute(reduce, work)


def _hip_pack_rows2(x, att, ids, out):
    check(lib().gnnagg_pack_rows2(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(att.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                  int(ids.numel()), int(x.shape[1]), int(att.shape[1]), ctypes.c_void_p(out.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def _hip_unpack_rows2(buf, n, x_out, att_out):
    check(lib().gnnagg_unpack_rows2(ctypes.c_void_p(buf.data_ptr()), int(n), int(x_out.shape[1]), int(att_out.shape[1]),
                                    ctypes.c_void_p(x_out.data_ptr()), ctypes.c_void_p(att_out.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


class PartitionedGAT:
    """Row-partitioned fused GAT.  The attention terms of a halo row travel WITH its feature row: ONE exchange of rows
    [x | att] (feat + 2 heads floats) per step.  overlap=True (balanced mode): the local-source edges' numerators and denominators
    are computed while the exchange is in flight, the halo-source pass adds its own and divides (gnnagg_gat_run_part); both passes
    index X_ext = [X_local ; X_halo] and att_ext slots.  row_slice / bounds / num_cols / transport as in PartitionedGCN; with the
    "rccl" transport, or on a single rank, the whole step is ONE C-ABI call (gnnagg_dist_step_gat)."""

    def __init__(self, ptr, idx, feat=256, heads=8, group=None, device=None, mode="balanced", rank=None, world=None,
                 offline=False, row_slice=False, bounds=None, num_cols=None, transport="torch", overlap=True, pack_fn2=None,
                 unpack_fn2=None, build_aggregators=True, stages=1):
        """pack_fn2 / unpack_fn2 / build_aggregators=False: test doubles for the CPU tests of the exchange (the product's pack,
        unpack and aggregation are HIP kernels without a CPU fallback)."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        aw = 2 * heads
        # the two-pass form runs on 16-byte lanes over one column tile (gnnagg_gat_run_part)
        self.overlap = bool(overlap) and mode == "balanced" and feat % 4 == 0 and (feat // heads) % 4 == 0 and feat <= 256
        self.hx = HaloExchange(ptr, idx, rank=rank, world=world, group=group, device=device, offline=offline,
                               row_slice=row_slice, bounds=bounds, num_cols=num_cols, transport=transport,
                               stages=stages if self.overlap else 1, row_bytes=4 * (feat + aw))
        hx = self.hx
        self.feat, self.heads, self.mode = feat, heads, mode
        self.pack_fn2, self.unpack_fn2 = pack_fn2 or _hip_pack_rows2, unpack_fn2 or _hip_unpack_rows2
        self.x_ext = hx.alloc_x_ext(feat)
        self.att_ext = hx.alloc_x_ext(aw)
        self.y = torch.empty((hx.n_local, feat), dtype=torch.float32, device=device)
        self.den = torch.empty((max(hx.n_local, 1), heads), dtype=torch.float32, device=device)
        self.send_buf = torch.empty((max(hx.n_send, 1), feat + aw), dtype=torch.float32, device=device)
        self.recv_buf = torch.empty((max(hx.n_halo, 1), feat + aw), dtype=torch.float32, device=device)
        self._step = ctypes.c_int64(0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        if not build_aggregators:
            return
        if self.overlap:
            pl, il, pr, ir, _ = hx.split_local_remote()
            self.agg_loc = Aggregator_GAT(t(pl), t(il), feat, feat)
            # one aggregator per stage (X_ext slots, like agg_loc); the LAST stage's pass divides every row, so it always exists
            self.agg_rem_stages = []
            stage_csr = hx.split_remote_stages()
            for st, (ps_, is_, _) in enumerate(stage_csr):
                if len(is_) == 0 and st + 1 < len(stage_csr):
                    self.agg_rem_stages.append(None)
                else:
                    self.agg_rem_stages.append(Aggregator_GAT(t(ps_), t((is_ + hx.n_local).astype(np.int32)), feat, feat))
            self.agg_rem = self.agg_rem_stages[0] if hx.n_stages == 1 else None
            if device.type == "cuda" and not offline and (hx.rccl is not None or hx.world == 1):
                n, S = hx.world, hx.n_stages
                sc = (ctypes.c_longlong * (n * S))(*[int(v) for v in hx.stage_send.reshape(-1)])
                rc = (ctypes.c_longlong * (n * S))(*[int(v) for v in hx.stage_recv.reshape(-1)])
                hs = (ctypes.c_int64 * S)(*[(a._h.value if a is not None else 0) for a in self.agg_rem_stages])
                check(lib().gnnagg_dist_step_create_staged(hx.rccl._h if hx.rccl is not None else ctypes.c_int64(0), self.agg_loc._h, S,
                                                           hs, ctypes.c_void_p(hx.send_ids.data_ptr()) if hx.n_send else None, sc, rc,
                                                           ctypes.byref(self._step)))
        else:
            self.d_ptr = t(hx.local_ptr)
            self.d_idx = t(hx.local_idx)
            self.agg = Aggregator_GAT(self.d_ptr, self.d_idx, feat, feat)
            if mode == "balanced":
                self.agg.schedule_balanced(0)

    def close(self):
        if self._step.value:
            lib().gnnagg_dist_step_destroy(self._step)
            self._step = ctypes.c_int64(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_local(self, x_local, att_local):
        n = self.hx.n_local
        self.x_ext[:n].copy_(x_local)
        self.att_ext[:n].copy_(att_local.reshape(n, -1))

    def exchange(self, async_op=False):
        """The all-to-all(s) of [x | att] rows into recv_buf -- one per stage; returns the work handle (None: nothing to wait for).
        The received rows still have to be split into the halo tails (finish_exchange)."""
        hx = self.hx
        if hx.offline or (hx.world == 1 and not (dist.is_available() and dist.is_initialized())):
            return None
        n = hx.n_local
        if hx.n_send:
            self.pack_fn2(self.x_ext[:n], self.att_ext[:n], hx.send_ids, self.send_buf)
        works = [hx.exchange_stage(st, self.send_buf[:max(hx.n_send, 1)], self.recv_buf[:max(hx.n_halo, 1)], async_op) for st in range(hx.n_stages)]
        return _StagedWork(works) if async_op else None

    def finish_exchange(self, stage=None):
        """recv_buf rows -> the halo tails of x_ext / att_ext (all stages, or one)"""
        hx = self.hx
        if not hx.n_halo or hx.offline:
            return
        r0, r1 = (0, hx.n_halo) if stage is None else (int(hx.stage_recv0[stage]), int(hx.stage_recv0[stage + 1]))
        if r1 > r0:
            self.unpack_fn2(self.recv_buf[r0:r1], r1 - r0, self.x_ext[hx.n_local + r0:hx.n_local + r1], self.att_ext[hx.n_local + r0:hx.n_local + r1])

    def compute(self, slope=0.2, work=None):
        """The aggregation kernels of one step (the halo tails of x_ext / att_ext are filled by `work` + finish_exchange, or by
        hand in the single-process tests)."""
        if self.overlap:
            self.agg_loc.run_part(self.x_ext, self.att_ext, self.y, self.den, 1, self.heads, slope)   # overlaps the all-to-all
            last = max(st for st, a in enumerate(self.agg_rem_stages) if a is not None)
            for st, a in enumerate(self.agg_rem_stages):
                if work is not None:
                    work.stage(st).wait()
                    self.finish_exchange(st)
                if a is not None:
                    a.run_part(self.x_ext, self.att_ext, self.y, self.den, 2 if st == last else 3, self.heads, slope)
        else:
            if work is not None:
                work.wait()
                self.finish_exchange()
            # the kernel indexes att by X_ext slot; only the first n_local rows are destinations
            self.agg.run_with_feat(self.x_ext, self.att_ext, self.y, 128, self.mode, self.feat, self.heads, slope)
        return self.y

    def step(self, slope=0.2):
        if self._step.value:   # ONE host call (gnnagg_dist_step_gat)
            hx = self.hx
            check(lib().gnnagg_dist_step_gat(self._step, ctypes.c_void_p(self.x_ext.data_ptr()), ctypes.c_void_p(self.att_ext.data_ptr()),
                                             int(hx.n_local), ctypes.c_void_p(self.send_buf.data_ptr()),
                                             ctypes.c_void_p(self.recv_buf.data_ptr()), ctypes.c_void_p(self.den.data_ptr()),
                                             ctypes.c_void_p(self.y.data_ptr()), int(self.feat), int(self.heads), ctypes.c_float(slope),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return self.y
        work = self.exchange(async_op=self.overlap)
        if work is None:
            self.finish_exchange()
        return self.compute(slope, work)

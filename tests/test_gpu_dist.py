"""GPU, 2 processes sharing cuda:0, gloo transport: the complete row-partitioned step (plan exchange, HIP pack kernel,
asynchronous all-to-all, local-source aggregation overlapped with it, halo-source aggregation with the accumulate
flag) end to end.  RCCL needs one GPU per rank, so on the single-GPU test box the collective itself runs over gloo
(which moves CUDA tensors through the host); everything else is the production path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_computing_amd as gnc
        from gnn_computing_amd.dist import PartitionedGCN
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        V, E, F = 6000, 150000, 128
        ptr_t, idx_t = gnc.graph.powerlaw_csr(V, E, seed=7)
        ptr, idx = ptr_t.numpy(), idx_t.numpy()
        rng = np.random.default_rng(3)
        x, val = rng.standard_normal((V, F), dtype=np.float32), rng.standard_normal(E, dtype=np.float32)
        pg = PartitionedGCN(ptr, idx, val, F, device="cuda:0", overlap=overlap)
        r0, r1 = int(pg.hx.bounds[rank]), int(pg.hx.bounds[rank + 1])
        pg.set_local_x(torch.from_numpy(x[r0:r1]).cuda())
        ok = True
        for _ in range(3):  # steady state: buffers are reused step after step
            y = pg.step().cpu().numpy()
            ref = orc.gcn_seq(ptr, idx, val, x)[r0:r1]
            scale = orc.gcn_abs_scale(ptr, idx, val, x)[r0:r1]
            ok = ok and bool(np.all(np.abs(y - ref) <= 1e-5 * scale + 1e-30))
        ok_halo = np.array_equal(pg.x_halo.cpu().numpy(), x[pg.hx.halo_ids])
        q.put((rank, ok, ok_halo, pg.hx.n_halo, pg.hx.n_send))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_one_gpu(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, ok_halo, n_halo, n_send in res:
        assert ok, "rank %d: partitioned aggregation differs from the single-GPU oracle result" % rank
        assert ok_halo, "rank %d: halo rows differ" % rank
        assert n_halo > 0 and n_send > 0


def test_rccl_transport_behind_the_cabi_single_rank(tmp_path):
    """gnnagg_dist_*: librccl is loaded on demand, a communicator is created on this GPU from a unique id (directly and
    through the id file a C++ launcher uses), the self part of an all-to-all-v is a stream-ordered copy, a halo exchange
    with nothing to send is a no-op, handles are checked.  (One GPU: the peer-to-peer part needs one GPU per rank.)"""
    import ctypes
    import sys
    sys.path.insert(0, ROOT)
    import gnn_computing_amd as gnc
    from gnn_computing_amd import _lib
    L = gnc.lib()
    torch.cuda.set_device(0)
    buf = ctypes.create_string_buffer(128)
    _lib.check(L.gnnagg_dist_unique_id(buf))
    assert any(b != 0 for b in buf.raw)
    comm = ctypes.c_int64(0)
    _lib.check(L.gnnagg_dist_comm_create(buf, 0, 1, ctypes.byref(comm)))
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.check(L.gnnagg_dist_comm_info(comm, ctypes.byref(r), ctypes.byref(w)))
    assert (r.value, w.value) == (0, 1)
    src = torch.randn(1000, device="cuda:0")
    dst = torch.zeros(1000, device="cuda:0")
    cnt = (ctypes.c_longlong * 1)(1000)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    _lib.check(L.gnnagg_dist_alltoallv(comm, src.data_ptr(), cnt, dst.data_ptr(), cnt, 4, st.cuda_stream))
    st.synchronize()
    assert torch.equal(src, dst)
    bad = (ctypes.c_longlong * 1)(999)
    assert L.gnnagg_dist_alltoallv(comm, src.data_ptr(), cnt, dst.data_ptr(), bad, 4, None) == _lib.ERR_ARG
    zero = (ctypes.c_longlong * 1)(0)
    x = torch.randn((10, 8), device="cuda:0")
    _lib.check(L.gnnagg_dist_halo_exchange(comm, x.data_ptr(), None, zero, zero, 8, None, None, None))
    _lib.check(L.gnnagg_dist_comm_destroy(comm))
    assert L.gnnagg_dist_comm_destroy(comm) == _lib.ERR_ARG
    comm2 = ctypes.c_int64(0)
    path = str(tmp_path / "id.bin")
    open(path, "wb").write(b"\x07" * 128)   # what a crashed earlier launch left: rank 0 replaces it
    _lib.check(L.gnnagg_dist_comm_create_from_file(path.encode(), 0, 1, 5, ctypes.byref(comm2)))
    rec = open(path, "rb").read()
    assert len(rec) == 16 + 128 and rec[:8] == b"GNNAGID1" and rec[16:] != b"\x07" * 128   # {magic, world, id}; no tokens at world 1
    _lib.check(L.gnnagg_dist_comm_destroy(comm2))
    assert L.gnnagg_dist_comm_create(buf, 3, 2, ctypes.byref(comm2)) == _lib.ERR_ARG


def test_dist_step_driver_single_rank(tmp_path):
    """drivers/dist_step.cpp: the multi-process C++ driver of the row-partitioned step, started as the only rank."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import gnn_computing_amd as gnc
    d = str(tmp_path) + "/"
    ptr, idx = gnc.graph.powerlaw_csr(5000, 60000, seed=3)
    gnc.graph.write_graph_files(d, "tiny", ptr.numpy(), idx.numpy(), text=True)
    exe = os.path.join(ROOT, "drivers", "dist_step.out")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "drivers")])
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, "--dataset", "tiny", "--datadir", d, "--feature-len", "64", "--iters", "5", "--idfile", d + "id"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[0]["n_local"] == 5000 and lines[0]["n_halo"] == 0 and lines[0]["seconds"] > 0
    assert lines[1]["summary"] == "slowest rank" and lines[1]["edges_per_s"] > 0

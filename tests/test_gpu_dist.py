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

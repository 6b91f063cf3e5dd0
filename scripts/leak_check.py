import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_computing_amd as gnc
dev = torch.device("cuda", 0)
ptr, idx = gnc.graph.powerlaw_csr(20000, 400000, seed=1, device=dev)
x = torch.randn((20000, 64), device=dev); y = torch.empty_like(x)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for i in range(300):
    a = gnc.Aggregator_GCN(ptr, idx, None, 64, 64)
    a.run(x, y, 128, "balanced"); a.run(x, y, 128, 0)
    a.schedule(gnc.Schedule.neighbor_grouping, [8]); a.run(x, y, 128, 1)
    g = gnc.Aggregator_GAT(ptr, idx, 64, 64)
    att = torch.randn((20000, 2), device=dev)
    g.run(x, att, y, 128, "balanced")
    del a, g
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB, after %.1f MB, delta %.1f MB" % (free0/1e6, free1/1e6, (free0-free1)/1e6))

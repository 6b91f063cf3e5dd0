"""gnn_computing_amd -- MI355X-native neighbor aggregation (GCN/GraphSAGE SpMM, GAT edge-softmax
SDDMM + SpMM) behind the reference's Aggregator API.  All compute is in libgnnagg.so (HIP, gfx950);
this package is the host-side mirror of the reference's operator interface plus input generators.
"""
from . import _lib
from ._lib import GnnAggError, lib  # noqa: F401
from .aggregator import (  # noqa: F401
    Aggregator, Aggregator_GAT, Aggregator_GCN, Schedule,
    gat_init, gat_run, gat_run_add_to_center, gat_run_div_each, gat_run_u_add_v, gat_schedule,
    gcn_init, gcn_run, gcn_schedule, gcn_update_val, new_load, load_graph_host,
    reorder_csr, neighbor_grouping_schedule, locality_schedule, partition_rows, halo_plan, cluster_reorder, matmul_NN,
)
from . import graph  # noqa: F401
from . import probe  # noqa: F401
# (gnn_computing_amd.extras -- torch.autograd wrappers over the backward entry points -- needs libgnnagg_extras.so: out of scope per
# SURVEY 2.2, not imported here)

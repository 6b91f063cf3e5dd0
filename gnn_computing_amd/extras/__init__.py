"""Out-of-scope extras, kept out of the default package and the default library: torch.autograd
wrappers over the backward entry points gnnagg_gcn_run_bwd / gnnagg_gat_run_bwd.  They exist only in libgnnagg_extras.so
(`make -C gnn_computing_amd/csrc extras`; load it with GNNAGG_LIB=<repo>/gnn_computing_amd/libgnnagg_extras.so)."""
from .. import _lib

if not _lib.has_extras():
    raise ImportError("gnn_computing_amd.extras needs libgnnagg_extras.so (make -C gnn_computing_amd/csrc extras; "
                      "GNNAGG_LIB=.../libgnnagg_extras.so): the shipped libgnnagg.so has no backward entry points")

from . import autograd  # noqa: E402,F401
from .autograd import gat_aggregate, gcn_aggregate  # noqa: E402,F401

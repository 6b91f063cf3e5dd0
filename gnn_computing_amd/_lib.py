"""ctypes binding of libgnnagg.so (include/gnnagg.h).

There is no CPU fallback: if the HIP library is missing this module raises at import of the
symbol table, and every compute entry point returns GNNAGG_ERR_HIP when no GPU is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNNAGG_LIB", os.path.join(_HERE, "libgnnagg.so"))  # override: A/B builds in scripts/

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_IO = 0, 1, 2, 3, 4
SCHED_LOCALITY, SCHED_NEIGHBOR_GROUPING, SCHED_LOCALITY_NEIGHBOR_GROUPING, SCHED_NOP = 0, 1, 2, 3
REDUCE_SUM, REDUCE_MEAN, REDUCE_MAX = 0, 1, 2
MODE_ROWS, MODE_SCHEDULED, MODE_BALANCED = 0, 1, 2
FLAG_ACCUMULATE = 1
FLAG_RELU = 2

c_int, c_float, c_void_p, c_char_p, c_int64 = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p,
                                                ctypes.c_int64)
P_INT = ctypes.POINTER(ctypes.c_int)
PP_INT = ctypes.POINTER(P_INT)

# name -> (restype, argtypes); every symbol include/gnnagg.h declares
SIGNATURES = {
    "gnnagg_last_error": (c_char_p, []),
    "gnnagg_version": (c_int, []),
    "gnnagg_set_abort_on_error": (None, [c_int]),
    # A: flat API (reference Figure7/kernel.cpp:15-35)
    "GCN_init_impl": (c_int64, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "GCN_update_val_impl": (None, [c_int64, c_void_p]),
    "GCN_run_impl": (None, [c_int64, c_void_p, c_void_p, c_int, c_int, c_int]),
    "GCN_schedule_impl": (None, [c_int64, P_INT]),
    "GAT_init_impl": (c_int64, [c_void_p, c_void_p, c_int, c_int]),
    "GAT_run_impl": (None, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "GAT_run_u_add_v_impl": (None, [c_int64, c_void_p, c_void_p, c_int]),
    "GAT_run_add_to_center_impl": (None, [c_int64, c_void_p, c_void_p, c_int]),
    "GAT_run_div_each_impl": (None, [c_int64, c_void_p, c_void_p, c_int]),
    "GAT_schedule_impl": (None, [c_int64, P_INT]),
    # B
    "gnnagg_gcn_create": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_int64)]),
    "gnnagg_gat_create": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_int64)]),
    "gnnagg_destroy": (c_int, [c_int64]),
    "gnnagg_set_stream": (c_int, [c_int64, c_void_p]),
    "gnnagg_set_option": (c_int, [c_int64, c_char_p, c_int]),
    "gnnagg_plan_info": (c_int, [c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong),
                                 ctypes.POINTER(ctypes.c_longlong)]),
    "gnnagg_update_val": (c_int, [c_int64, c_void_p]),
    "gnnagg_set_row_aux": (c_int, [c_int64, c_void_p]),
    "gnnagg_schedule": (c_int, [c_int64, c_int, P_INT, c_int]),
    "gnnagg_schedule_balanced": (c_int, [c_int64, c_int]),
    "gnnagg_balanced_params": (c_int, [c_int64, P_INT, P_INT]),
    "gnnagg_balanced_partitions": (c_int, [c_int64, P_INT, P_INT]),
    "gnnagg_rows_blocked_ranges": (c_int, [c_int64, P_INT]),
    "gnnagg_mode_params": (c_int, [c_int64, c_int, P_INT, P_INT]),
    "gnnagg_num_target": (c_int, [c_int64, c_int, P_INT]),
    "gnnagg_get_schedule": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnnagg_gcn_run": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_int]),
    "gnnagg_gcn_run_ex": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int]),
    "gnnagg_gcn_probe_gather": (c_int, [c_int64, c_void_p, c_int, c_int]),
    "gnnagg_gat_probe_gather": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_int]),
    "gnnagg_probe_row_gather": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, ctypes.c_longlong, c_int, c_void_p]),
    "gnnagg_gcn_run_clock": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, P_INT, P_INT]),
    "gnnagg_wall_clock_hz": (ctypes.c_longlong, []),
    "gnnagg_gcn_run_edgewise": (c_int, [c_int64, c_void_p, c_void_p, c_int]),
    "gnnagg_check_csr": (c_int, [c_int64, c_int, P_INT, P_INT]),
    "gnnagg_csr2edgelist": (c_int, [c_int64, c_void_p]),
    "gnnagg_matmul_nn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gnnagg_gcn_run_with_nn": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "gnnagg_gat_run": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "gnnagg_gat_run_part": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p]),
    "gnnagg_gat_run_att": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_float]),
    "gnnagg_gat_run_u_add_v": (c_int, [c_int64, c_void_p, c_void_p]),
    "gnnagg_gat_run_add_to_center": (c_int, [c_int64, c_void_p, c_void_p]),
    "gnnagg_gat_run_div_each": (c_int, [c_int64, c_void_p, c_void_p]),
    "gnnagg_spmm_naive": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gnnagg_validate": (c_int, [c_void_p, c_void_p, c_int, P_INT, c_void_p]),
    "gnnagg_validate_reordered": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, P_INT, c_void_p]),
    # C
    "gnnagg_load_graph": (c_int, [c_char_p, c_char_p, c_char_p, c_int, P_INT, P_INT, PP_INT, PP_INT, PP_INT, PP_INT]),
    "gnnagg_free_host": (None, [c_void_p]),
    "gnnagg_reorder_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gnnagg_neighbor_grouping_schedule": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, P_INT]),
    "gnnagg_locality_schedule": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, P_INT]),
    "gnnagg_cluster_reorder": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_int, ctypes.c_ulonglong, c_void_p, P_INT]),
    "gnnagg_cluster_reorder_ex": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_int, ctypes.c_ulonglong, c_int, c_int, c_void_p, P_INT]),
    # D
    "gnnagg_partition_rows": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "gnnagg_halo_plan": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, PP_INT,
                                 c_void_p, P_INT]),
    "gnnagg_halo_stage_plan": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, P_INT, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnnagg_pack_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gnnagg_halo_plan_slice": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, PP_INT,
                                       c_void_p, P_INT]),
    "gnnagg_dist_unique_id": (c_int, [c_void_p]),
    "gnnagg_dist_comm_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int64)]),
    "gnnagg_dist_comm_create_from_file": (c_int, [c_char_p, c_int, c_int, c_int, ctypes.POINTER(c_int64)]),
    "gnnagg_dist_comm_destroy": (c_int, [c_int64]),
    "gnnagg_dist_comm_info": (c_int, [c_int64, P_INT, P_INT]),
    "gnnagg_dist_transport_info": (c_int, [c_int64, c_char_p, c_int, c_char_p, c_int, P_INT]),
    "gnnagg_dist_alltoallv": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gnnagg_dist_halo_exchange": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gnnagg_pack_rows2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gnnagg_unpack_rows2": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gnnagg_dist_step_create": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    "gnnagg_dist_step_create_staged": (c_int, [c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    "gnnagg_dist_step_info": (c_int, [c_int64, P_INT, P_INT]),
    "gnnagg_dist_step_destroy": (c_int, [c_int64]),
    "gnnagg_dist_step_gcn": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gnnagg_dist_step_gat": (c_int, [c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                     c_void_p]),
}

# Section E of the header: only libgnnagg_extras.so (-DGNNAGG_EXTRAS; GNNAGG_LIB=.../libgnnagg_extras.so) exports these
EXTRA_SIGNATURES = {
    "gnnagg_gcn_run_bwd": (c_int, [c_int64, c_void_p, c_void_p, c_int]),
    "gnnagg_gat_run_bwd": (c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int]),
}

_lib = None
_has_extras = False


class GnnAggError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gnnagg error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Loads libgnnagg.so and types every entry point.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libgnnagg.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C gnn_computing_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        global _has_extras
        _has_extras = all(hasattr(L, name) for name in EXTRA_SIGNATURES)
        if _has_extras:
            for name, (res, args) in EXTRA_SIGNATURES.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        L.gnnagg_set_abort_on_error(0)  # Python callers get exceptions, not exit(1)
        _lib = L
    return _lib


def has_extras():
    """True when the loaded library is libgnnagg_extras.so (backward entry points, the older kernel forms): second-tier tests only."""
    lib()
    return _has_extras


def check(rc):
    if rc != OK:
        raise GnnAggError(rc, lib().gnnagg_last_error().decode())

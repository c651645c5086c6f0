"""ctypes binding of libpa_hip.so (the C ABI of include/pa_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no CPU or PyTorch fallback for the device path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# dmabuf IPC is the only mode this driver stack supports: hipIpcGetMemHandle (the ipc push transport, RCCL's own p2p set-up) fails
# with "invalid argument" without it.  Must be in the environment before the HIP runtime initialises; harmless for one process.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# torch is imported first on purpose: it ships its own libamdhip64.so.7 / librccl.so.1 and the
# dynamic loader then resolves libpa_hip.so's dependencies to those already-loaded copies
# (one HIP runtime per process).  torch is plumbing here (torch.distributed bootstrap), not compute.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PA_HIP_LIBRARY") or os.path.join(_HERE, "libpa_hip.so")   # the override: probe builds of the library


class PAError(RuntimeError):
    """Non-zero status from libpa_hip (the reference raises a Julia exception at the same places)."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C partitionedarrays.jl_amd/csrc`). The HIP library is required; there is no fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

P = C.c_void_p
PP = C.POINTER(C.c_void_p)
i32, i64, f64 = C.c_int32, C.c_int64, C.c_double
cint = C.c_int

lib.pa_last_error.restype = C.c_char_p
lib.pa_last_error.argtypes = []
lib.pa_version.restype = cint

_SIGS = {
    "pa_device_count": [C.POINTER(cint)],
    "pa_ctx_create": [cint, PP],
    "pa_ctx_destroy": [P],
    "pa_ctx_sync": [P],
    "pa_ctx_stream": [P, cint, PP],
    "pa_ctx_device_info": [P, C.POINTER(cint), C.POINTER(cint), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t],
    "pa_ctx_stream_priority": [P, cint, C.POINTER(cint), C.POINTER(cint), C.POINTER(cint)],
    "pa_event_create": [P, PP],
    "pa_event_destroy": [P],
    "pa_event_record": [P, cint],
    "pa_event_elapsed_ms": [P, P, C.POINTER(C.c_float)],
    "pa_vec_create": [P, i64, i64, PP],
    "pa_vec_wrap": [P, P, i64, i64, PP],
    "pa_vec_destroy": [P],
    "pa_vec_sizes": [P, C.POINTER(i64), C.POINTER(i64)],
    "pa_vec_data": [P, PP],
    "pa_vec_upload": [P, P, i64, i64],
    "pa_vec_download": [P, P, i64, i64],
    "pa_vec_fill": [P, cint, f64],
    "pa_vec_copy": [P, P, cint],
    "pa_vec_axpby": [P, f64, P, f64, cint],
    "pa_vec_dot": [P, P, C.POINTER(f64)],
    "pa_vec_dot_result": [P, PP],
    "pa_ctx_read_scalar": [P, C.POINTER(f64)],
    "pa_graph_begin": [P],
    "pa_graph_end": [P, PP],
    "pa_graph_launch": [P],
    "pa_graph_destroy": [P],
    "pa_matrix_create": [P, P, P, P, PP],
    "pa_matrix_destroy": [P],
    "pa_mul": [P, P, P, P],
    "pa_mul5": [P, P, P, P, f64, f64],
    "pa_mul_no_lat": [P, P, P, P],
    "pa_mul_all": [P, i32, P, P, f64, f64],
    "pa_matrix_ghost_from_buffer": [P, C.POINTER(cint)],
    "pa_mul_dot": [P, P, P, P, cint, cint],
    "pa_mul_all_dot": [P, i32, P, P, cint],
    "pa_vec_dot_slot": [P, P, cint, cint],
    "pa_vec_axpby_slot": [P, f64, cint, cint, P, f64, cint, cint, cint],
    "pa_cg_update": [P, P, P, P, cint, cint, cint, cint],
    "pa_cg_r_update": [P, P, cint, cint, cint, cint],
    "pa_cg_xu_update": [P, P, P, cint, cint, cint, cint],
    "pa_ctx_slot_ptr": [P, cint, PP],
    "pa_ctx_write_slot": [P, cint, f64],
    "pa_ctx_read_slots": [P, cint, cint, C.POINTER(f64)],
    "pa_csr_create": [P, i64, i64, i64, P, P, cint, cint, P, PP],
    "pa_csr_value_dict": [P, C.POINTER(cint)],
    "pa_csr_device_bytes": [P, C.POINTER(i64)],
    "pa_csr_stream_bytes": [P, C.POINTER(i64)],
    "pa_ctx_arena_build": [P],
    "pa_ctx_arena_hint": [P, cint],
    "pa_ctx_arena_release": [P],
    "pa_ctx_arena_info": [P, C.POINTER(i64), C.POINTER(cint), C.POINTER(i64), C.POINTER(i64), C.POINTER(f64), C.POINTER(cint)],
    "pa_ctx_arena_map": [P, C.POINTER(i64), P, i64, C.POINTER(i64)],
    "pa_ctx_arena_stats": [P] + [C.POINTER(i64)] * 8,
    "pa_csr_debug_array": [P, cint, P, i64, C.POINTER(i64)],
    "pa_ctx_keep_raw_columns": [P, cint],
    "pa_csr_has_raw_columns": [P, C.POINTER(cint)],
    "pa_csr_drop_raw_columns": [P],
    "pa_csr_select_rows": [P, P, P, C.c_int32, P],
    "pa_csr_select_rows_lower": [P, i64, P, C.c_int32, P],
    "pa_csr_diagonal": [P, P],
    "pa_gs_create_from_blocks": [P, P, cint, C.POINTER(P)],
    "pa_csr_greedy_coloring": [P, P, C.POINTER(C.c_int32)],
    "pa_csr_greedy_coloring_by_levels": [P, P, P, C.POINTER(i32)],
    "pa_csr_color_affinity": [P, P, C.c_int32, P, i64, P],
    "pa_hpcg_own_block_create": [P] + [i64] * 9 + [C.POINTER(P), P],
    "pa_hpcg_rhs": [P] + [i64] * 9 + [P],
    "pa_coo_assemble": [P, i64, P, P, P, C.c_int32, P, P, P, P, P, P, i64, P, cint, C.POINTER(P)],
    "pa_coo_assembly_info": [P] + [C.POINTER(i64)] * 5 + [C.POINTER(f64)],
    "pa_coo_subassemble": [P, i64, P, P, P, C.c_int32, P, P, P, P, P, P, C.POINTER(P)],
    "pa_coo_subassembly_info": [P] + [C.POINTER(i64)] * 4,
    "pa_coo_subassembly_ghost_rows": [P, P, P, P, P],
    "pa_coo_assemble_finish": [P, i64, P, P, P, C.POINTER(P)],
    "pa_coo_assembly_ghosts": [P, P],
    "pa_coo_assembly_blocks": [P, C.POINTER(P), C.POINTER(P)],
    "pa_coo_assembly_download": [P, cint, P, P, P],
    "pa_coo_assembly_destroy": [P],
    "pa_csr_memory_class": [P, C.POINTER(cint)],
    "pa_vec_memory_class": [P, C.POINTER(cint)],
    "pa_csr_create_mixed": [P, i64, i64, i64, P, cint, P, cint, cint, P, PP],
    "pa_csr_create_from_csc": [P, i64, i64, i64, P, P, cint, cint, P, PP],
    "pa_csr_set_alpha_inside": [P, cint],
    "pa_csr_update_values": [P, P],
    "pa_csr_update_values_from": [P, P, i64],
    "pa_csr_destroy": [P],
    "pa_gs_create": [P, i64, i64, i64, P, P, P, cint, cint, PP],
    "pa_gs_destroy": [P],
    "pa_gs_info": [P, C.POINTER(i64), C.POINTER(i64)],
    "pa_gs_sweep": [P, P, P, cint, cint],
    "pa_host_greedy_coloring": [i64, P, P, cint, P, C.POINTER(i32)],
    "pa_host_remap_int32": [P, i64, P, i32],
    "pa_rowset_create": [P, i64, P, cint, PP],
    "pa_rowset_destroy": [P],
    "pa_gs_color_update": [P, P, P, P, P],
    "pa_gs_color_sweep": [P, cint, P, P, P, cint],
    "pa_gs_color_symmetric_sweep": [P, cint, P, P, P, cint],
    "pa_gs_color_symmetric_sweep_zero": [P, P, cint, P, P, P],
    "pa_transfer_create": [P, i64, P, cint, PP],
    "pa_transfer_destroy": [P],
    "pa_transfer_attach_rows": [P, P],
    "pa_transfer_restrict_fused": [P, P, P, P],
    "pa_transfer_restrict": [P, P, P, P],
    "pa_transfer_prolongate": [P, P, P],
    "pa_scatter_create": [P, i64, i64, P, cint, PP],
    "pa_scatter_destroy": [P],
    "pa_scatter_add": [P, P, P, cint],
    "pa_csr_info": [P] + [C.POINTER(i64)] * 6,
    "pa_csr_encoding": [P] + [C.POINTER(i64)] * 3,
    "pa_csr_xwin_info": [P] + [C.POINTER(i64)] * 4,
    "pa_csr_xring_info": [P, C.POINTER(i64)],
    "pa_spmv": [P, P, cint, P, cint, f64, f64],
    "pa_spmv_tune_output": [P, P, cint, P, cint, cint, i32, P, P, C.POINTER(i32), C.POINTER(i32)],
    "pa_ctx_pci_bus_id": [P, C.c_char_p, C.c_size_t],
    "pa_csr_create_transpose": [P, PP],
    "pa_matrix_create_transposed": [P, P, P, P, PP],
    "pa_mul5_transpose": [P, P, P, P, f64, f64],
    "pa_mul5_transpose_all": [P, i32, P, P, f64, f64],
    "pa_csr_download_entries": [P, P, P],
    "pa_matrix_fused": [P, C.POINTER(cint), C.POINTER(i64)],
    "pa_ctx_reload_env": [P],
    "pa_ctx_fused_launches": [P, C.POINTER(i64), C.POINTER(i64)],
    "pa_csr_locality_order": [P, P, C.POINTER(i64), C.POINTER(i64)],
    "pa_csr_create_permuted": [P, P, P, PP],
    "pa_csr_create_transpose_ranked": [P, P, PP],
    "pa_csr_create_colsplit": [P, cint, PP],
    "pa_csr_create_empty": [P, i64, i64, PP],
    "pa_csr_chain_info": [P, C.POINTER(C.c_int32), C.POINTER(i64)],
    "pa_coo_keep_input_slots": [P, cint],
    "pa_coo_reuse_scatter": [P, P, P, PP, i64, P],
    "pa_scatter_download": [P, P],
    "pa_sell_create": [P, i64, i64, i64, P, P, cint, cint, P, cint, PP],
    "pa_sell_destroy": [P],
    "pa_sell_info": [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
    "pa_sell_spmv": [P, P, cint, P, cint, f64, f64],
    "pa_vec32_create": [P, i64, i64, PP],
    "pa_vec32_destroy": [P],
    "pa_vec32_upload": [P, P, i64, i64],
    "pa_vec32_download": [P, P, i64, i64],
    "pa_vec32_fill": [P, cint, C.c_float],
    "pa_csr32_create": [P, i64, i64, i64, P, P, cint, cint, P, PP],
    "pa_csr32_create_from_csc": [P, i64, i64, i64, P, P, cint, cint, P, PP],
    "pa_csr32_destroy": [P],
    "pa_csr32_info": [P, C.POINTER(cint), C.POINTER(i64), C.POINTER(i64)],
    "pa_spmv32": [P, P, cint, P, cint, C.c_float, C.c_float],
    "pa_csr_pell_info": [P, C.POINTER(cint), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(cint)],
    "pa_fem_triplets_device": [P, C.c_int32, P, P, P, P, C.POINTER(i64), C.POINTER(P), C.POINTER(P), C.POINTER(P)],
    "pa_triplets_free": [P, P, P, P],
    "pa_triplets_download": [P, P, i64, P],
    "pa_comm_create_all": [P, C.c_int32, P],
    "pa_exchange_rccl_all": [P, P, C.c_int32, cint],
    "pa_vec32_data": [P, C.POINTER(P)],
    "pa_exchange_pack_raw": [P, P, i64, cint, cint],
    "pa_exchange_finish_raw": [P, P, i64, i64, cint, cint],
    "pa_exchange_pack32": [P, P, cint],
    "pa_exchange_finish32": [P, P, cint],
    "pa_csr_pell_lean_info": [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
    "pa_plan_create": [P, i32, i64, i32, P, P, P, i32, P, P, P, cint, PP],
    "pa_plan_destroy": [P],
    "pa_plan_buffers": [P, cint, PP, C.POINTER(i64), PP, C.POINTER(i64)],
    "pa_exchange_pack": [P, P, cint],
    "pa_exchange_finish": [P, P, cint],
    "pa_exchange_local": [PP, i32, cint],
    "pa_exchange_rccl": [P, P, cint],
    "pa_exchange_push_local": [PP, i32, PP, cint],
    "pa_exchange_finish_all": [PP, i32, PP, cint],
    "pa_plan_ipc_blob_size": [P, C.POINTER(i64)],
    "pa_plan_ipc_blob": [P, P, i64],
    "pa_plan_ipc_connect": [P, i32, P, P],
    "pa_plan_ipc_status": [P, C.POINTER(cint)],
    "pa_exchange_push_ipc": [P, P, cint],
    "pa_comm_unique_id": [C.c_char_p],
    "pa_comm_create": [P, C.c_char_p, cint, cint, PP],
    "pa_comm_destroy": [P],
    "pa_comm_allreduce_sum": [P, P, i64, cint],
    "pa_comm_barrier": [P],
    "pa_comm_info": [P, C.POINTER(cint), C.POINTER(cint)],
    "pa_host_hpcg_build_matrix": [i64] * 9 + [P, P, P, P, P, C.POINTER(i64)],
    "pa_host_laplacian_fdm": [i32, P, P, P, P, P, P, C.POINTER(i64)],
    "pa_host_laplacian_fem": [i32, P, P, P, P, P, P, P, C.POINTER(i64)],
    "pa_host_find_owner_block": [i32, P, P, PP, P, i64, P],
    "pa_host_filter_ghost": [i32, P, P, i64, P, i64, P, P, C.POINTER(i64)],
    "pa_host_global_to_local_block": [i32, P, P, P, P, i64, P, i64, P],
    "pa_host_compresscoo_csr": [P, P, P, i64, i64, i64, cint, P, P, P, C.POINTER(i64)],
    "pa_host_split_csr": [i64, i64, i64, P, P, P, P, P, P, P, P, P, C.POINTER(i64), C.POINTER(i64)],
    "pa_host_check_spmv_encodings": [i64, i64, i64, P, P, cint] + [C.POINTER(i64)] * 4,
    "pa_host_check_xw_groups": [i64, i64, i64, P, P, cint] + [C.POINTER(i64)] * 5,
    "pa_host_hpcg_ghosts": [i64] * 9 + [P, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
    "pa_host_hpcg_split_csr": [i64] * 9 + [P, i64, P, P, P, P, P, P, P],
    "pa_host_hpcg_ghost_block": [i64] * 9 + [P, i64, P, P, P],
    "pa_host_color_rowptrs": [i64, P, P, P, C.c_int32, P],
    "pa_host_color_split": [i64, i64, P, P, P, P, P, P, P, i32, P, P, P, P],
    "pa_host_hpcg_split_csr64": [i64] * 9 + [P, i64, P, P, P, P, P, P, P],
}
# every symbol the header declares (tests/test_abi.py checks this list against include/pa_hip.h)
EXPORTS = ["pa_version", "pa_last_error"] + list(_SIGS)

for _name, _args in _SIGS.items():
    _f = getattr(lib, _name)          # AttributeError here == the library does not export the symbol
    _f.argtypes = _args
    _f.restype = cint

SEG_OWN, SEG_GHOST, SEG_LOCAL = 0, 1, 2
CONSISTENT, ASSEMBLE = 0, 1
STREAM_COMPUTE, STREAM_COMM = 0, 1
N_SLOTS, SLOT_ONE = 16, -1
UNIQUE_ID_BYTES = 128


def check(status: int):
    if status != 0:
        raise PAError(f"libpa_hip status {status}: {lib.pa_last_error().decode()}")


def call(name: str, *args):
    check(getattr(lib, name)(*args))


def ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags.c_contiguous, "need a C-contiguous numpy array"
    return a.ctypes.data_as(P)

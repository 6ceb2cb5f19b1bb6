"""ctypes binding of the C ABI declared in include/latticenet_hip.h.

The library is the product: if it is missing this module raises — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads torch's libamdhip64 first so both share ONE HIP runtime by SONAME)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LATTICE_NET_LIB") or os.path.join(_PKG, "liblatticenet_hip.so")  # override: experimental builds (tools)

LN_STATUS_TABLE_FULL = 1
LN_STATUS_KEY_RANGE = 2
LN_STATUS_BUCKET_OVERFLOW = 4
LN_BUILD_WRITE_IDX = 1
LN_BUILD_CLEAR_FIRST = 2
LN_BUILD_ATOMIC_PATH = 4
LN_BUILD_CANONICAL_ROWS = 8
LN_BUILD_SORTED_CSR = 16
LN_NOT_VISITED = -2
LN_CONV_FLIP_NEIGHBOURS = 1
LN_CONV_TRANSPOSED_FILTER = 2
LN_CONV_BANK_READY = 4
LN_MAX_POS_DIM = 6
LN_KEYS_RAW = 0
LN_KEYS_LATTICE = 1
LN_XCD_GROUPS = 8
LN_SLOT_MAP_INTS = 32


class LnTable(C.Structure):
    """Mirror of `struct LnTable` (include/latticenet_hip.h)."""

    _fields_ = [
        ("capacity", C.c_int),
        ("pos_dim", C.c_int),
        ("slot_keys", C.c_void_p),
        ("slot_tok", C.c_void_p),
        ("slot_cnt", C.c_void_p),
        ("entries", C.c_void_p),
        ("keys", C.c_void_p),
        ("nr_filled", C.c_void_p),
        ("status", C.c_void_p),
        ("host_counters", C.c_void_p),
        ("host_seq", C.c_int),
        ("key_format", C.c_int),
        ("row_limit", C.c_int),
        ("slot_map", C.c_void_p),
        ("bucket_slots_max", C.c_int),
        ("batch_points", C.c_int),
        ("batch_key_step", C.c_int),
        ("row_regions", C.c_void_p),
    ]


class LnCsr(C.Structure):
    """Mirror of `struct LnCsr` (include/latticenet_hip.h)."""

    _fields_ = [
        ("grp_start", C.c_void_p),
        ("csr_tok", C.c_void_p),
        ("seg_desc", C.c_void_p),
        ("seg_count", C.c_void_p),
        ("seg_region", C.c_longlong),
        ("planes", C.c_void_p),
        ("dense", C.c_int),
    ]


class LatticeNetHipError(RuntimeError):
    pass


_vp, _i, _ll, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_size_t
_T = C.POINTER(LnTable)
_CSR = C.POINTER(LnCsr)

# name -> (restype, argtypes).  Every symbol the header declares is listed here; tests check the
# library exports each of them.
SIGNATURES = {
    "ln_last_error_string": (C.c_char_p, []),
    "ln_version": (C.c_char_p, []),
    "ln_abi_hash": (C.c_char_p, []),
    "ln_kernel_names": (C.c_char_p, []),
    "ln_profile_begin": (_i, [C.c_char_p, _i]),
    "ln_profile_end": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "ln_profile_end_table": (_i, [C.c_char_p, _i]),
    "ln_table_clear": (_i, [_T, _vp, _ll, _vp]),
    "ln_build_workspace_bytes": (_sz, [_ll, _i]),
    "ln_table_bucket_count": (_i, [_i]),
    "ln_build_concurrency": (_i, [_i]),
    "ln_build_splat": (_i, [_T, _vp, _vp, _i, _vp, _vp, _i, _CSR, _vp, _sz, _vp, _ll, _vp]),
    "ln_rehash": (_i, [_T, _vp]),
    "ln_canonicalize": (_i, [_T, _vp, _ll, _CSR, _vp, _sz, _vp]),
    "ln_splat_accumulate": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ln_csr_workspace_bytes": (_sz, [_ll, _i]),
    "ln_csr_max_segments": (_ll, [_ll, _i]),
    "ln_csr_build": (_i, [_vp, _ll, _i, _CSR, _vp, _sz, _vp]),
    "ln_csr_reduce_rows": (_i, [_CSR, _vp, _ll, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_csr_sort": (_i, [_CSR, _i, _vp, _sz, _ll, _vp]),
    "ln_splat_accumulate_and_neighbours": (_i, [_CSR, _vp, _ll, _vp, _vp, _i, _i, _i, _vp, _T, _i, _vp, _vp]),
    "ln_csr_segment_max": (_i, [_CSR, _vp, _ll, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ln_csr_group_sizes": (_i, [_CSR, _vp, _i, _i, _vp, _vp]),
    "ln_distribute": (_i, [_T, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _CSR, _vp, _sz, _vp, _ll, _vp]),
    "ln_coarsen": (_i, [_T, _i, _T, _CSR, _vp, _sz, _vp]),
    "ln_neighbours": (_i, [_T, _i, _T, _i, _i, _i, _i, _vp, _vp]),
    "ln_im2row": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_im2rowindices": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ln_row2im": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_conv_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ln_conv_forward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ln_conv_bank_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ln_conv_forward_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ln_conv_grad_filter_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ln_conv_grad_filter": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ln_slice_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_slice_forward_prepare_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _ll, _vp]),
    "ln_slice_forward_ordered": (_i, [_T, _CSR, _vp, _vp, _vp, _i, _i, _vp, _vp, _ll, _vp]),
    "ln_conv_row_partition": (_i, [_vp]),
    "ln_slice_no_precomputation": (_i, [_T, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ln_slice_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_gather_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_gather_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_slice_classify_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ln_csr_reduce_rows_f16": (_i, [_CSR, _vp, _ll, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_splat_accumulate_and_neighbours_f16": (_i, [_CSR, _vp, _ll, _vp, _vp, _i, _i, _i, _vp, _T, _i, _vp, _vp]),
    "ln_slice_forward_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ln_slice_forward_f16_prepare_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _ll, _vp]),
    "ln_conv_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ln_conv_forward_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ln_conv_grad_filter_f16_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ln_conv_grad_filter_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ln_linear_act_forward": (_i, [_vp, _vp, _vp, _ll, _i, _i, C.c_float, _vp, _vp]),
    "ln_linear_act_backward_workspace_bytes": (_sz, [_i, _i]),
    "ln_linear_act_backward": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, C.c_float, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ln_arena_init": (_i, [_vp, _ll, _ll, _ll, _vp]),
    "ln_weight_norm_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ln_weight_norm_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ln_distribute_centre": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp]),
    "ln_pointnet_reduce_workspace_bytes": (_sz, [_i, _i]),
    "ln_pointnet_reduce_forward": (_i, [_CSR, _vp, _ll, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ln_pointnet_reduce_backward": (_i, [_vp, _i, _vp, _vp, _ll, _i, _vp, _vp]),
    "ln_linear_backward_workspace_bytes": (_sz, [_i, _i, _i]),
    "ln_linear_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ln_nll_workspace_bytes": (_sz, []),
    "ln_nll_forward": (_i, [_vp, _vp, _ll, _i, _ll, _vp, _sz, _vp, _vp]),
    "ln_nll_backward": (_i, [_vp, _vp, _vp, _ll, _i, _ll, _vp, _vp]),
    "ln_max_centre_forward": (_i, [_vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _vp]),
    "ln_max_centre_backward_workspace_bytes": (_sz, [_ll, _i, _i]),
    "ln_max_centre_backward": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ln_group_norm_workspace_bytes": (_sz, [_i]),
    "ln_group_norm_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ln_group_norm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "ln_group_norm_forward_rows": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "ln_group_norm_backward_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "ln_slice_classify_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ln_slice_classify_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


def load():
    """Loads liblatticenet_hip.so; raises loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LatticeNetHipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python lattice_net_amd/build_ext.py`). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the .so is git-ignored: make sure it was built from the header these SIGNATURES were written against
    from . import build_ext
    try:
        want = build_ext.abi_hash()
    except OSError:
        want = None  # header not shipped next to the package: nothing to compare with
    have = lib.ln_abi_hash().decode()
    if want is not None and have != want:
        raise LatticeNetHipError(f"{LIB_PATH} was built from a different include/latticenet_hip.h (library {have}, header {want}): "
                                 "rebuild it with `python lattice_net_amd/build_ext.py`")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ln_last_error_string().decode("utf-8", "replace")
        raise LatticeNetHipError(f"{what or 'latticenet_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor as an int (None -> NULL); ctypes converts it to void*."""
    return None if t is None else t.data_ptr()


def stream_ptr(device) -> int:
    """Raw hipStream_t of torch's current stream on `device` (fast path: no Stream object is built)."""
    cur = torch.cuda.current_device()
    idx = device.index if device.index is not None else cur
    if idx != cur:
        # the C ABI launches on the raw stream it is handed and never selects a device itself: a stream of another
        # device than the current one would be used under the wrong device context
        raise LatticeNetHipError(f"lattice tensors live on cuda:{idx} but the current device is cuda:{cur}; wrap the call in "
                                 f"`with torch.cuda.device({idx}):` (or torch.cuda.set_device) — one process drives one GPU")
    return torch._C._cuda_getCurrentRawStream(idx)


def host_floats(values):
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    return arr

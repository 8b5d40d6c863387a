"""ctypes binding of ``librgcn_hip.so`` (C ABI declared in ``include/rgcn_hip.h``).

There is no CPU fallback: if the library is missing, or was built against another
ABI version, every compute entry point raises.  ``import torch`` must precede the
``dlopen`` so that the HIP runtime already in the process (torch's bundled
``libamdhip64.so.7``) satisfies the library's NEEDED entry by soname.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, POINTER, c_char_p, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (loads the HIP runtime first)

ABI_VERSION = 24
LIB_NAME = "librgcn_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

RGCN_OK = 0
RGCN_ERR_ARG = -1
RGCN_ERR_RANGE = -2
RGCN_ERR_HIP = -3
RGCN_ERR_UNSUPPORTED = -4
RGCN_ERR_WORKSPACE = -5

_P = c_void_p      # device pointers travel as integers (tensor.data_ptr())
_I64 = c_int64

class SlabJob(ctypes.Structure):
    """``rgcn_slab_job``: a pending parameter-gradient slab reduction (plain device pointers)."""
    _fields_ = [("slab", c_void_p), ("bias_part", c_void_p), ("splits", ctypes.c_int32), ("K1", ctypes.c_int32),
                ("Kc", ctypes.c_int32), ("N", ctypes.c_int32), ("grad_weight", c_void_p), ("grad_root", c_void_p),
                ("grad_bias", c_void_p)]


class SeqArg(ctypes.Structure):
    """``rgcn_seq_arg``"""
    _fields_ = [("kind", ctypes.c_int32), ("index", ctypes.c_int32), ("value", ctypes.c_int64)]


class SeqCall(ctypes.Structure):
    """``rgcn_seq_call``"""
    _fields_ = [("fn", ctypes.c_int32), ("num_args", ctypes.c_int32), ("first_arg", ctypes.c_int64)]


SEQ_IMM, SEQ_FLOAT, SEQ_BASE, SEQ_JOB, SEQ_STREAM, SEQ_ARRAY = range(6)
# entry points rgcn_sequence_run can forward to (RGCN_FN_* of the header, in its order)
SEQ_FUNCTIONS = ("rgcn_absmax", "rgcn_absmax_multi", "rgcn_absmax_pack", "rgcn_weights_split_pack_multi", "rgcn_aggregate",
                 "rgcn_aggregate_and_reduce", "rgcn_aggregate_amax", "rgcn_aggregate_deferred", "rgcn_transform_fwd_split",
                 "rgcn_transform_bwd_input_split", "rgcn_transform_first_split", "rgcn_transform_bwd_params_split_begin",
                 "rgcn_slab_reduce", "rgcn_layer_fwd_fused", "rgcn_layer_bwd_input_fused",
                 "rgcn_transform_bwd_input_chain_split")
# their HOST array parameters: position -> (entries are device pointers?, position of the parameter holding the count)
SEQ_HOST_ARRAYS = {
    "rgcn_absmax_multi": {1: (True, 0), 2: (False, 0), 3: (True, 0)},
    "rgcn_absmax_pack": {6: (True, 5), 7: (True, 5), 8: (False, 5), 9: (False, 5), 10: (False, 5), 11: (True, 5), 12: (False, 5)},
    "rgcn_weights_split_pack_multi": {1: (True, 0), 2: (True, 0), 3: (False, 0), 4: (False, 0), 5: (False, 0), 6: (True, 0),
                                      7: (True, 0), 8: (True, 0), 9: (False, 0)},
}


# name -> (restype, argtypes); mirrors include/rgcn_hip.h one to one
PROTOTYPES = {
    "rgcn_abi_version": (c_int, []),
    "rgcn_strerror": (c_char_p, [c_int]),
    "rgcn_graph_create": (c_int, [_P, _P, _I64, _I64, _I64, _P, POINTER(c_void_p)]),
    "rgcn_graph_create_bipartite": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, POINTER(c_void_p)]),
    "rgcn_graph_destroy": (None, [c_void_p]),
    "rgcn_graph_num_edges": (_I64, [c_void_p]),
    "rgcn_graph_num_nodes": (_I64, [c_void_p]),
    "rgcn_graph_num_relations": (_I64, [c_void_p]),
    "rgcn_graph_num_levels": (c_int, [c_void_p, c_int]),
    "rgcn_graph_weight_bound": (c_float, [c_void_p, c_int]),
    "rgcn_graph_arrays": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                  POINTER(c_void_p), POINTER(c_void_p)]),
    "rgcn_graph_export": (c_int, [c_void_p, c_int, _P, _P, _P, _P, _P]),
    "rgcn_graph_import": (c_int, [_I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, POINTER(c_void_p)]),
    "rgcn_aggregate_workspace_bytes": (c_size_t, [c_void_p, c_int, _I64]),
    "rgcn_aggregate": (c_int, [c_void_p, c_int, _P, _I64, _P, _P, c_size_t, _P]),
    "rgcn_aggregate_f16": (c_int, [c_void_p, c_int, _P, _I64, _P, _P, c_size_t, _P]),
    "rgcn_aggregate_level": (c_int, [c_void_p, c_int, c_int, _P, _I64, _P, _P, c_size_t, _P, _P]),
    "rgcn_graph_tile_mask": (c_void_p, [c_void_p, c_int, POINTER(c_int64)]),
    "rgcn_transform_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "rgcn_transform_fwd_f16_workspace_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_transform_fwd_f16": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _I64, _I64, _I64, _I64, _P, _P, c_size_t, _P]),
    "rgcn_transform_bwd_input": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "rgcn_transform_bwd_params_workspace_bytes": (c_size_t, [_I64, _I64, _I64, _I64]),
    "rgcn_transform_bwd_params": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P,
                                          c_size_t, _P]),
    "rgcn_transform_bwd_params_begin": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, c_size_t, _P,
                                                POINTER(SlabJob)]),
    "rgcn_slab_reduce": (c_int, [POINTER(SlabJob), _P]),
    "rgcn_aggregate_and_reduce": (c_int, [c_void_p, c_int, _P, _I64, _P, _P, c_size_t, POINTER(SlabJob), _P]),
    "rgcn_aggregate_amax": (c_int, [c_void_p, c_int, _P, _I64, _P, _P, c_size_t, POINTER(SlabJob), _P, _P]),
    "rgcn_absmax": (c_int, [_P, _I64, _P, _P, c_int, _P]),
    "rgcn_absmax_multi": (c_int, [c_int, _P, _P, _P, _P, c_int, _P]),
    "rgcn_absmax_pack": (c_int, [_P, _I64, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rgcn_weights_split_pack_multi": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "rgcn_weights_split_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_weights_split_pack": (c_int, [_P, _P, _I64, _I64, _I64, _P, c_size_t, _P]),
    "rgcn_transform_split_workspace_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_transform_fwd_split": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _I64, _I64, _I64, _I64, _P, c_float, _P, c_int,
                                         _P, _P, _P, c_size_t, _P, c_void_p, c_int, _P]),
    "rgcn_transform_bwd_input_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, c_float, _P, c_int,
                                               _P, _P, _P, c_size_t, _P, c_void_p, c_int, _P, c_float]),
    "rgcn_transform_bwd_input_chain_supported": (c_int, [_I64, _I64, _I64, _I64, _I64]),
    "rgcn_transform_bwd_input_chain_split": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, c_float, _P, _P,
                                                     _P, _P, c_size_t, _P, c_void_p, c_int, _P, c_float, _P, c_int, _I64,
                                                     _I64, _P]),
    "rgcn_aggregate_deferrable": (c_int, [c_void_p, c_int, _I64]),
    "rgcn_aggregate_deferred": (c_int, [c_void_p, c_int, _P, _I64, _P, _P, c_size_t, POINTER(SlabJob), _P]),
    "rgcn_transform_first_split": (c_int, [_P, _P, c_int, _I64, _I64, _I64, _I64, _P, c_int, _P, _P, c_size_t, _P]),
    "rgcn_transform_bwd_params_split_workspace_bytes": (c_size_t, [_I64, _I64, _I64, _I64]),
    "rgcn_transform_bwd_params_split_begin": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, c_float, _P, _P, c_int,
                                                      _P, _P, _P, _P, c_size_t, _P, POINTER(SlabJob)]),
    "rgcn_layer_fwd_fused_supported": (c_int, [_I64, _I64, _I64]),
    "rgcn_layer_fwd_fused": (c_int, [_P, _P, _P, _I64, _I64, _P, _P, _P, c_int, _P, c_int, _I64, _I64, _P, _P, _P, _P, _P]),
    "rgcn_layer_bwd_input_fused_supported": (c_int, [_I64, _I64, _I64]),
    "rgcn_layer_bwd_input_fused": (c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, c_int, _P, _I64, _I64, _P, c_float, _P, _P,
                                           _P, c_float]),
    "rgcn_sequence_run": (c_int, [_P, c_int, _P, _I64, _P, c_int, _P]),
    "distmult_fwd": (c_int, [_P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _P]),
    "rgcn_index_error_fetch": (c_int, [POINTER(c_int), _P]),
    "distmult_bwd_workspace_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_segment_sum_workspace_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_segment_sum": (c_int, [_P, _P, _I64, _I64, _I64, _P, _P, c_size_t, _P]),
    "distmult_rank_tails": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P, _P]),
    "rgcn_basis_compose": (c_int, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "rgcn_basis_compose_bwd_workspace_bytes": (c_size_t, [_I64, _I64, _I64]),
    "rgcn_basis_compose_bwd": (c_int, [_P, _P, _P, _I64, _I64, _I64, _P, _P, _P, c_size_t, _P]),
    "distmult_bce_reduce": (c_int, [_P, _P, _P, _I64, _P, _P, _P, _P, _I64, _P]),
    "distmult_score_all_tails": (c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _I64, _P, _P, _P]),
    "distmult_bwd": (c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "rgcn_adam_workspace_bytes": (c_size_t, [c_int, _P]),
    "rgcn_adam_clip_step": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int,
                                    c_float, _P, _P, _P, c_size_t, _P]),
    "rgcn_sample_batch": (c_int, [_P, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "distmult_bce_fwd": (c_int, [_P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _P, _I64, _I64, _P, _P, _P]),
    "distmult_bce_bwd": (c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P, _P, _P, _P,
                                 c_size_t, c_int, _P]),
}

_lib = None


class RGCNLibraryError(RuntimeError):
    """The HIP library is missing / unloadable / of the wrong ABI."""


def load() -> ctypes.CDLL:
    """dlopen the library once and type every entry point.  Raises loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RGCNLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C primekg_rgcn_linkprediction_amd/csrc`). There is no "
            f"CPU/PyTorch fallback for the R-GCN path.")
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise RGCNLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise RGCNLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.rgcn_abi_version()
    if got != ABI_VERSION:
        raise RGCNLibraryError(f"{LIB_PATH} has ABI version {got}, host code expects {ABI_VERSION}")
    _lib = lib
    return lib


def available() -> bool:
    try:
        load()
        return True
    except RGCNLibraryError:
        return False


def strerror(code: int) -> str:
    return load().rgcn_strerror(code).decode()


def check(code: int, what: str) -> None:
    """Map a C return code to the Python exception the reference's stack would raise."""
    if code == RGCN_OK:
        return
    msg = f"{what}: {strerror(code)} (code {code})"
    if code == RGCN_ERR_RANGE:
        raise IndexError(msg)
    if code in (RGCN_ERR_ARG, RGCN_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)

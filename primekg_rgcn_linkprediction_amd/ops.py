"""Tensor-level wrappers over the C ABI (``include/rgcn_hip.h``).

These do the checks PyG/torch would do on the host (dtype, device, contiguity,
shape), hand raw device pointers + the current HIP stream to the library and
return torch-allocated outputs.  No layer arithmetic happens in Python (the only
torch ops here verify a persisted bucketing against the columns it claims to
describe), and there is no CPU path: a CPU tensor raises.

Reference call sites replaced: ``RGCNConv.forward`` as used at
``src/models/rgcn.py:123,128`` and ``LinkPredictor.forward`` (``rgcn.py:189-213``).
"""
from __future__ import annotations

import ctypes
import os
import pickle
from collections import OrderedDict
from typing import Optional, Tuple

import torch

from . import _lib


# ----------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_REC = None           # the Region recorder while a pass is being sized / recorded (see Region below), else None


def _empty(*shape, dtype, device) -> torch.Tensor:
    """``torch.empty`` for every tensor a pass allocates: while a Region records the pass, the allocation is part of
    the record (sizes first, then views of the pass's arena)"""
    if _REC is None:
        return torch.empty(*shape, dtype=dtype, device=device)
    return _REC.alloc(shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape, dtype, device)


def _L():
    """the library - or, while a Region records a pass, the proxy that notes every call"""
    return _lib.load() if _REC is None or _REC.proxy is None else _REC.proxy


def _stream() -> int:
    """the current HIP stream of the current device as an integer handle.  The raw binding is ~30x
    cheaper than ``torch.cuda.current_stream().cuda_stream`` (11 us of Python per launch measured)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device):
    """Device guard for a launch: a real ``torch.cuda.device`` switch only when ``device`` is not
    already current (the usual case - one process per GPU - costs a comparison instead of a
    Python-level guard push/pop around each of the ~14 launches of a step)."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_gpu(name: str, t: torch.Tensor, dtype: torch.dtype) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(
            f"{name} is on {t.device}: the R-GCN engine runs on MI355X (HIP) tensors only; "
            f"there is no CPU fallback in this package")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


# Arithmetic of the three dense transforms (rows A6 / A7).  "split" (default): fp32 values carried as
# fp16 hi/lo pairs on the fp16 matrix cores, fp32 accumulate (csrc/rgcn_transform_split.hip; ~2^-22 per
# product, inside the north star's 1e-5 gate on every config); "fp32": v_mfma_f32_32x32x2_f32, bit for
# bit an fmaf chain.  Shapes the split kernels do not tile run the fp32 kernels either way.
GEMM_PRECISION = os.environ.get("RGCN_GEMM_PRECISION", "split")
if GEMM_PRECISION not in ("split", "fp32"):
    raise ValueError(f"RGCN_GEMM_PRECISION must be 'split' or 'fp32', got {GEMM_PRECISION!r}")


def _use_split(precision: Optional[str], k_dim: int, multiple: int) -> int:
    """0: the fp32 kernels; 1: split precision (three fp16 MFMA passes); 2: ``"half"`` - one pass, operands
    rounded to fp16 under their per-tensor scales, fp32 accumulate (BASELINE configs[4]'s gradient GEMMs).
    Widths the split kernels do not tile run the fp32 kernels (more precise, never less)."""
    mode = GEMM_PRECISION if precision is None else precision
    if mode not in ("split", "fp32", "half"):
        raise ValueError(f"precision must be 'split', 'half' or 'fp32', got {mode!r}")
    if mode == "fp32" or k_dim % multiple:
        return 0
    return 2 if mode == "half" else 1


# An "amax buffer" (include/rgcn_hip.h, RGCN_AMAX_FLOATS): AMAX_FLOATS floats; its VALUE, max |tensor|, is the
# maximum over its 256 heads (every 8th entry; the rest is never touched).  Producing kernels publish wave maxima
# into 64 of the heads with an atomic max on the bit pattern (spread so the atomics of a launch do not queue on
# one address), so the heads must be zero before a producer runs; ``absmax`` clears buffers on the side.
AMAX_FLOATS = 2048
AMAX_HEAD_STRIDE = 8


def amax_buffer(device, count: int = 1) -> torch.Tensor:
    """``count`` zeroed amax buffers as one ``[count, AMAX_FLOATS]`` tensor (one fill launch)"""
    return torch.zeros(count, AMAX_FLOATS, dtype=torch.float32, device=device)


def amax_value(buf: torch.Tensor) -> torch.Tensor:
    """the number an amax buffer stands for (a 0-dim device tensor)"""
    return buf.view(-1)[::AMAX_HEAD_STRIDE].max()


def _check_amax(name: str, t: Optional[torch.Tensor], device) -> None:
    if t is None:
        return
    _need_gpu(name, t, torch.float32)
    if t.numel() != AMAX_FLOATS or t.device != device:
        raise ValueError(f"{name} must be an amax buffer ({AMAX_FLOATS} floats) on {device}")


# Operand maxima that somebody else already knows (round 4): the optimizer is the only writer of the embedding table and
# of the layers' weights, and ``adam_clip_step(amax_out=)`` leaves max |param| of what it just wrote in an amax buffer.
# ``set_amax_hint(param, buffer)`` says so; ``amax_hint(param)`` returns the buffer while the tensor has not been
# modified through torch since (``_version``) - the encoder's first launch then only splits the weights under the
# given maxima instead of scanning the 8 MB table and the weights at the head of the step's latency chain.
_AMAX_HINTS = {}


def set_amax_hint(t: torch.Tensor, buf: Optional[torch.Tensor]) -> None:
    import weakref
    if buf is None:
        _AMAX_HINTS.pop(id(t), None)
        return
    _check_amax("buf", buf, t.device)
    key = id(t)
    _AMAX_HINTS[key] = (weakref.ref(t, lambda _, k=key: _AMAX_HINTS.pop(k, None)), buf, t._version)


def amax_hint(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    h = _AMAX_HINTS.get(id(t))
    if h is None or h[0]() is not t or h[2] != t._version or not t.is_contiguous() or t.dtype != torch.float32:
        return None
    return h[1]


def absmax(x: torch.Tensor, out: Optional[torch.Tensor] = None, clear: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``max |x|`` as an amax buffer (``rgcn_absmax``; ``amax_value`` reads it): the operand scale of the
    split-precision transforms for a tensor no kernel of this library produced (the embedding table,
    the incoming gradient).  One launch, no atomics; ``clear`` (``[k, AMAX_FLOATS]``, contiguous): amax
    buffers whose heads the same launch zeroes - the ones the kernels of the coming pass publish into."""
    _need_gpu("x", x, torch.float32)
    lib = _L()
    with _on(x.device):
        if out is None:
            out = _empty(AMAX_FLOATS, dtype=torch.float32, device=x.device)
        _check_amax("out", out, x.device)
        count = 0
        if clear is not None:
            _need_gpu("clear", clear, torch.float32)
            if clear.numel() % AMAX_FLOATS or clear.device != x.device:
                raise ValueError("clear must hold whole amax buffers on x's device")
            count = clear.numel() // AMAX_FLOATS
        rc = lib.rgcn_absmax(_ptr(x), x.numel(), _ptr(out), _ptr(clear), count, _stream())
    _lib.check(rc, "rgcn_absmax")
    return out


def absmax_many(tensors, outs, clear: Optional[torch.Tensor] = None) -> None:
    """``absmax`` of up to 8 tensors in ONE launch (``rgcn_absmax_multi``), tensor i into amax buffer ``outs[i]``:
    the first launch of a pass - the embedding table and both layers' weights."""
    if not tensors or len(tensors) != len(outs) or len(tensors) > 8:
        raise ValueError("1..8 tensors, one amax buffer each")
    dev = tensors[0].device
    for i, (t, o) in enumerate(zip(tensors, outs)):
        _need_gpu(f"tensors[{i}]", t, torch.float32)
        _check_amax(f"outs[{i}]", o, dev)
    count = 0
    if clear is not None:
        _need_gpu("clear", clear, torch.float32)
        if clear.numel() % AMAX_FLOATS or clear.device != dev:
            raise ValueError("clear must hold whole amax buffers on the tensors' device")
        count = clear.numel() // AMAX_FLOATS
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)(*[_ptr(t) for t in tensors])
    numels = (ctypes.c_int64 * n)(*[t.numel() for t in tensors])
    outp = (ctypes.c_void_p * n)(*[_ptr(o) for o in outs])
    with _on(dev):
        rc = _L().rgcn_absmax_multi(n, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(numels, ctypes.c_void_p),
                                           ctypes.cast(outp, ctypes.c_void_p), _ptr(clear), count, _stream())
    _lib.check(rc, "rgcn_absmax_multi")


class SplitWeights:
    """``[W ; root]`` of one layer split into fp16 hi / lo images for the split-precision transforms
    (``rgcn_weights_split_pack``): made once per step by ``split_weights`` and handed to
    ``transform_fwd`` / ``transform_bwd_input`` as ``packed=`` so that neither splits them again."""

    def __init__(self, buf: torch.Tensor, weight: torch.Tensor, root: Optional[torch.Tensor]):
        self.buf, self.shape, self.has_root = buf, tuple(weight.shape), root is not None

    def matches(self, weight: torch.Tensor, root: Optional[torch.Tensor]) -> bool:
        return tuple(weight.shape) == self.shape and (root is not None) == self.has_root and weight.device == self.buf.device


def fused_supported(num_relations: int, d_in: int, d_out: int) -> bool:
    """does the one-kernel layer forward cover this shape (in split precision)?"""
    return GEMM_PRECISION != "fp32" and bool(_query("rgcn_layer_fwd_fused_supported", int(num_relations), int(d_in), int(d_out)))


def split_weights(weight: torch.Tensor, root: Optional[torch.Tensor]) -> Optional[SplitWeights]:
    """-> ``SplitWeights`` (None in fp32 mode or for widths the split kernels do not tile)"""
    return split_weights_many([(weight, root)])[0]


def split_weights_many(layers, amax=None, clear: Optional[torch.Tensor] = None):
    """``[(weight, root | None), ...]`` (up to 4 layers) -> ``[SplitWeights | None, ...]`` in ONE launch
    (``rgcn_weights_split_pack_multi``).  ``amax``: per layer ``(weight_amax, root_amax)`` buffers when the
    pass's first launch (``absmax_many``) - or the optimizer (``adam_clip_step(amax_out=)``) - already left the
    weights' maxima; otherwise the kernel scans them.  ``clear``: amax buffers whose heads the launch clears on the
    side (it can then be the first launch of a pass)."""
    if not layers or len(layers) > 4:
        raise ValueError("1..4 layers")
    todo, out = [], [None] * len(layers)
    for i, (weight, root) in enumerate(layers):
        _need_gpu("weight", weight, torch.float32)
        if weight.dim() != 3:
            raise ValueError("weight must be [R, d_in, d_out]")
        r, d_in, d_out = weight.shape
        if root is not None:
            _need_gpu("root", root, torch.float32)
            if tuple(root.shape) != (d_in, d_out):
                raise ValueError(f"root must be [{d_in}, {d_out}]")
        if GEMM_PRECISION != "fp32" and d_in % 32 == 0 and d_out % 32 == 0:
            todo.append(i)
    count = 0
    if clear is not None:
        _need_gpu("clear", clear, torch.float32)
        if clear.numel() % AMAX_FLOATS:
            raise ValueError("clear must hold whole amax buffers")
        count = clear.numel() // AMAX_FLOATS
    if not todo:
        if count:
            guard_torch_op("clearing amax buffers without a launch to carry it")
            clear.zero_()
        return out
    lib = _L()
    dev = layers[todo[0]][0].device
    n = len(todo)
    with _on(dev):
        sizes = [_query("rgcn_weights_split_bytes", *layers[i][0].shape) for i in todo]
        bufs = [_empty(sz, dtype=torch.uint8, device=dev) for sz in sizes]
        arr = ctypes.c_void_p * n
        i64 = ctypes.c_int64 * n
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)                                   # noqa: E731
        wam = ram = None
        if amax is not None:
            for i in todo:
                _check_amax("weight_amax", amax[i][0], dev)
                _check_amax("root_amax", amax[i][1], dev)
            wam = cast(arr(*[_ptr(amax[i][0]) for i in todo]))
            ram = cast(arr(*[_ptr(amax[i][1]) for i in todo]))
        rc = lib.rgcn_weights_split_pack_multi(
            n, cast(arr(*[_ptr(layers[i][0]) for i in todo])), cast(arr(*[_ptr(layers[i][1]) for i in todo])),
            cast(i64(*[layers[i][0].size(0) for i in todo])), cast(i64(*[layers[i][0].size(1) for i in todo])),
            cast(i64(*[layers[i][0].size(2) for i in todo])), wam, ram, cast(arr(*[_ptr(b) for b in bufs])),
            cast((ctypes.c_size_t * n)(*sizes)), _ptr(clear) if count else None, count, _stream())
    _lib.check(rc, "rgcn_weights_split_pack_multi")
    for i, b in zip(todo, bufs):
        out[i] = SplitWeights(b, layers[i][0], layers[i][1])
    return out


_MERGED_PACK_MAX = 1 << 17      # elements of [W ; root] up to which the merged first launch scans the weights per workgroup


def absmax_and_split(x: torch.Tensor, out: torch.Tensor, clear: Optional[torch.Tensor], layers):
    """The first launch of a forward pass as ONE launch (``rgcn_absmax_pack``): ``absmax(x, out, clear)`` and
    ``split_weights_many(layers)`` together -> ``[SplitWeights | None, ...]``.  Layers the split kernels do not
    tile (and fp32 mode) get None; if none is left this is ``absmax`` alone."""
    if not layers or len(layers) > 4:
        raise ValueError("1..4 layers")
    _need_gpu("x", x, torch.float32)
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    _check_amax("out", out, x.device)
    todo, packs = [], [None] * len(layers)
    for i, (weight, root) in enumerate(layers):
        _need_gpu("weight", weight, torch.float32)
        if weight.dim() != 3 or not weight.is_contiguous():
            raise ValueError("weight must be a contiguous [R, d_in, d_out]")
        r, d_in, d_out = weight.shape
        if root is not None:
            _need_gpu("root", root, torch.float32)
            if tuple(root.shape) != (d_in, d_out) or not root.is_contiguous():
                raise ValueError(f"root must be a contiguous [{d_in}, {d_out}]")
        if GEMM_PRECISION != "fp32" and d_in % 32 == 0 and d_out % 32 == 0:
            todo.append(i)
    if not todo:
        absmax(x, out, clear)
        return packs
    if max(layers[i][0].numel() + (layers[i][1].numel() if layers[i][1] is not None else 0) for i in todo) > _MERGED_PACK_MAX:
        # large weights (hidden 256: 1 MB per layer): in the merged launch every one of a layer's 64 workgroups
        # scans all of them for the maximum (35 us at C3); here one launch takes the maxima, a second one splits
        tensors, outs, wam = [x], [out], {}
        with _on(x.device):
            allbufs = _empty(2 * len(todo), AMAX_FLOATS, dtype=torch.float32, device=x.device)   # every head is written
        for k, i in enumerate(todo):
            bufs = allbufs[2 * k: 2 * k + 2]
            wam[i] = (bufs[0], bufs[1] if layers[i][1] is not None else None)
            tensors.append(layers[i][0])
            outs.append(bufs[0])
            if layers[i][1] is not None:
                tensors.append(layers[i][1])
                outs.append(bufs[1])
        absmax_many(tensors, outs, clear)
        sub = split_weights_many([layers[i] for i in todo], amax=[wam[i] for i in todo])
        for i, pk in zip(todo, sub):
            packs[i] = pk
        return packs
    count = 0
    if clear is not None:
        _need_gpu("clear", clear, torch.float32)
        if clear.numel() % AMAX_FLOATS or clear.device != x.device:
            raise ValueError("clear must hold whole amax buffers on x's device")
        count = clear.numel() // AMAX_FLOATS
    lib = _L()
    n = len(todo)
    with _on(x.device):
        sizes = [_query("rgcn_weights_split_bytes", *layers[i][0].shape) for i in todo]
        bufs = [_empty(sz, dtype=torch.uint8, device=x.device) for sz in sizes]
        arr, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)                                   # noqa: E731
        rc = lib.rgcn_absmax_pack(
            _ptr(x), x.numel(), _ptr(out), _ptr(clear), count, n,
            cast(arr(*[_ptr(layers[i][0]) for i in todo])), cast(arr(*[_ptr(layers[i][1]) for i in todo])),
            cast(i64(*[layers[i][0].size(0) for i in todo])), cast(i64(*[layers[i][0].size(1) for i in todo])),
            cast(i64(*[layers[i][0].size(2) for i in todo])), cast(arr(*[_ptr(b) for b in bufs])),
            cast((ctypes.c_size_t * n)(*sizes)), _stream())
    _lib.check(rc, "rgcn_absmax_pack")
    for i, b in zip(todo, bufs):
        packs[i] = SplitWeights(b, layers[i][0], layers[i][1])
    return packs


_QUERY_CACHE = {}


def _query(name: str, *args) -> int:
    """a pure size / capability query of the library, memoised (each ctypes call costs 2-4 us of host time)"""
    key = (name,) + args
    v = _QUERY_CACHE.get(key)
    if v is None:
        v = _QUERY_CACHE[key] = getattr(_lib.load(), name)(*args)
    return v


def _workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    if nbytes <= 0:
        return None
    return _empty(nbytes, dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------------
# bucketed graph (row A2) + cache
# ----------------------------------------------------------------------------------
class FusedPlan:
    """``BucketedGraph.fused_plan``: the CSR the one-kernel layer walks - ``rowptr`` int32[N * R + 1] / ``col`` int32
    (/ ``weight`` float32 for the transposed structure) in the structure's segment order, a segment longer than
    ``inline_limit`` edges replaced by ONE entry ``-(row + 1)`` (weight 1) naming its row of the pre-aggregated
    table - and ``hub``, the gather structure of those long segments (None if there is none)."""

    def __init__(self, inline_limit: int, rowptr, col, weight, hub, hub_rows: int, hub_edges: int):
        self.inline_limit, self.rowptr, self.col, self.weight, self.hub = inline_limit, rowptr, col, weight, hub
        self.hub_rows, self.hub_edges = hub_rows, hub_edges


class BucketedGraph:
    """Owner of one ``rgcn_graph`` handle: the CSR-by-relation structures (forward and
    transposed) of a static multigraph, built once on the device."""

    def __init__(self, edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int,
                 num_relations: int):
        self._handle = None
        if edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        if edge_type.dim() != 1 or edge_type.size(0) != edge_index.size(1):
            raise ValueError("edge_type must be [E] with E = edge_index.size(1)")
        _need_gpu("edge_index", edge_index, torch.int64)
        _need_gpu("edge_type", edge_type, torch.int64)
        if edge_type.device != edge_index.device:
            raise RuntimeError("edge_index and edge_type are on different devices")
        lib = _L()
        self.device = edge_index.device
        self.num_nodes, self.num_relations = int(num_nodes), int(num_relations)
        self.num_other_nodes = self.num_nodes      # rows of the gathered input
        self.bipartite = False
        self.weighted_shard = False
        self.num_edges = int(edge_index.size(1))
        handle = ctypes.c_void_p()
        with _on(self.device):
            rc = lib.rgcn_graph_create(_ptr(edge_index), _ptr(edge_type), self.num_edges,
                                       self.num_nodes, self.num_relations, _stream(),
                                       ctypes.byref(handle))
        _lib.check(rc, "rgcn_graph_create")
        self._handle = handle

    @classmethod
    def from_shard(cls, key_node: torch.Tensor, other_node: torch.Tensor, edge_type: torch.Tensor,
                   num_key_nodes: int, num_other_nodes: int, num_relations: int,
                   edge_weight: Optional[torch.Tensor] = None) -> "BucketedGraph":
        """One direction between two node sets: the shard a rank of a node-partitioned graph
        holds (``rgcn_graph_create_bipartite``).  Segments = (key_node, rel) over the rank's
        own ``num_key_nodes`` rows; ``other_node`` ids index the ``num_other_nodes`` gathered
        rows.  Without ``edge_weight``: mean mode; with it: weighted-sum mode."""
        self = cls.__new__(cls)
        self._handle = None
        for name, t in (("key_node", key_node), ("other_node", other_node), ("edge_type", edge_type)):
            _need_gpu(name, t, torch.int64)
            if t.dim() != 1 or t.size(0) != key_node.size(0):
                raise ValueError(f"{name} must be [E]")
        if edge_weight is not None:
            _need_gpu("edge_weight", edge_weight, torch.float32)
            if edge_weight.shape != key_node.shape:
                raise ValueError("edge_weight must be [E]")
        lib = _L()
        self.device = key_node.device
        self.num_nodes, self.num_relations = int(num_key_nodes), int(num_relations)
        self.num_other_nodes = int(num_other_nodes)
        self.bipartite = True
        self.weighted_shard = edge_weight is not None
        self.num_edges = int(key_node.size(0))
        handle = ctypes.c_void_p()
        with _on(self.device):
            rc = lib.rgcn_graph_create_bipartite(_ptr(key_node), _ptr(other_node), _ptr(edge_type),
                                                 self.num_edges, self.num_nodes, self.num_other_nodes,
                                                 self.num_relations, _ptr(edge_weight), _stream(),
                                                 ctypes.byref(handle))
        _lib.check(rc, "rgcn_graph_create_bipartite")
        self._handle = handle
        return self

    # -- persisted form (SURVEY section 8f "next" row 3) ------------------------------------
    SIDECAR_FORMAT = "rgcn-bucketed-v1"

    def state(self) -> dict:
        """The bucketed structure as a dict of CPU tensors (what ``save`` writes): both
        directions' ``rowptr / col / perm`` and ``cnt`` / ``w_t``."""
        if self.bipartite:
            raise ValueError("only whole graphs are persisted, not shard structures")
        fwd, bwd = self.arrays(False), self.arrays(True)
        names = ("rowptr", "col", "perm", "cnt"), ("rowptr_t", "col_t", "perm_t", "w_t")
        out = {"format": self.SIDECAR_FORMAT, "num_edges": self.num_edges, "num_nodes": self.num_nodes,
               "num_relations": self.num_relations}
        for keys, arrs in zip(names, (fwd, bwd)):
            out.update({k: a.cpu() for k, a in zip(keys, arrs)})
        return out

    @classmethod
    def from_state(cls, state: dict, device) -> "BucketedGraph":
        """Rebuild from ``state()`` on ``device`` without sorting (``rgcn_graph_import``); the
        index arrays are validated on the device (``IndexError`` if they are not a CSR of the
        stated sizes)."""
        if state.get("format") != cls.SIDECAR_FORMAT:
            raise ValueError(f"not a {cls.SIDECAR_FORMAT} file")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("BucketedGraph lives on an MI355X device; no CPU fallback")
        e, n, r = int(state["num_edges"]), int(state["num_nodes"]), int(state["num_relations"])
        want = {"rowptr": (torch.int32, n * r + 1), "col": (torch.int32, e), "perm": (torch.int64, e),
                "cnt": (torch.float32, n * r), "rowptr_t": (torch.int32, n * r + 1), "col_t": (torch.int32, e),
                "perm_t": (torch.int64, e), "w_t": (torch.float32, e)}
        dev = {}
        for k, (dt, size) in want.items():
            t = state[k]
            if t.dtype != dt or t.dim() != 1 or t.numel() != size:
                raise ValueError(f"{k} must be {dt}[{size}], got {t.dtype}{tuple(t.shape)}")
            dev[k] = t.to(device).contiguous()
        self = cls.__new__(cls)
        self._handle = None
        self.device = device
        self.num_nodes, self.num_relations, self.num_other_nodes, self.num_edges = n, r, n, e
        self.bipartite = self.weighted_shard = False
        handle = ctypes.c_void_p()
        with _on(device):
            rc = _L().rgcn_graph_import(e, n, r, *(_ptr(dev[k]) for k in want), _stream(),
                                               ctypes.byref(handle))
        _lib.check(rc, "rgcn_graph_import")
        self._handle = handle
        return self

    def save(self, path) -> None:
        torch.save(self.state(), path)

    @classmethod
    def load(cls, path, device) -> "BucketedGraph":
        return cls.from_state(torch.load(path, weights_only=True), device)

    @property
    def handle(self) -> ctypes.c_void_p:
        if self._handle is None:
            raise RuntimeError("BucketedGraph was destroyed")
        return self._handle

    def tile_mask_ptr(self, transposed: bool) -> Optional[int]:
        """device pointer (as int) of the relation-occupancy mask of 32-row tiles, or None"""
        if self.bipartite and transposed:
            raise ValueError("a shard structure has one direction only (transposed=False)")
        handle = self.handle                                         # raises once the graph is destroyed
        cache = self.__dict__.setdefault("_mask_ptrs", {})          # fixed for the life of the handle
        if transposed not in cache:
            cache[transposed] = _L().rgcn_graph_tile_mask(handle, int(transposed), None) or None
        return cache[transposed]

    def workspace_bytes(self, transposed: bool, d: int) -> int:
        """bytes of partial-sum workspace ``aggregate`` needs for rows of width d (memoised)"""
        handle = self.handle
        cache = self.__dict__.setdefault("_ws_bytes", {})
        key = (bool(transposed), int(d))
        if key not in cache:
            cache[key] = _L().rgcn_aggregate_workspace_bytes(handle, int(transposed), int(d))
        return cache[key]

    def num_levels(self, transposed: bool) -> int:
        handle = self.handle                                         # raises once the graph is destroyed
        cache = self.__dict__.setdefault("_levels", {})              # fixed for the life of the handle
        if transposed not in cache:
            cache[transposed] = _L().rgcn_graph_num_levels(handle, int(transposed))
        return cache[transposed]

    def deferrable(self, transposed: bool, d: int) -> bool:
        """may ``aggregate_deferred`` leave this direction's hub tails to the transform? (memoised)"""
        handle = self.handle
        cache = self.__dict__.setdefault("_deferrable", {})
        key = (bool(transposed), int(d))
        if key not in cache:
            cache[key] = bool(_L().rgcn_aggregate_deferrable(handle, int(transposed), int(d)))
        return cache[key]

    def weight_bound(self, transposed: bool) -> float:
        """``|aggregate(x) row| <= weight_bound * max |x|``: 1 for the mean structure, the largest
        per-segment sum of edge weights for a weighted one (memoised; fixed for the life of the handle)"""
        handle = self.handle
        cache = self.__dict__.setdefault("_wbound", {})
        key = bool(transposed and not self.bipartite)
        if key not in cache:
            cache[key] = float(_L().rgcn_graph_weight_bound(handle, int(key)))
        return cache[key]

    def arrays(self, transposed: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Copies of (rowptr int32[N*R+1], col int32[E], perm int64[E], val float32) on the
        device: val = cnt[N*R] (forward) or w_t[E] (transposed).  For parity tests."""
        lib = _L()
        nr, e = self.num_nodes * self.num_relations, self.num_edges
        with _on(self.device):
            rowptr = _empty(nr + 1, dtype=torch.int32, device=self.device)
            col = _empty(e, dtype=torch.int32, device=self.device)
            perm = _empty(e, dtype=torch.int64, device=self.device)
            val = _empty(e if transposed else nr, dtype=torch.float32, device=self.device)
            rc = lib.rgcn_graph_export(self.handle, int(transposed), _ptr(rowptr), _ptr(col), _ptr(perm),
                                       _ptr(val), _stream())
        _lib.check(rc, "rgcn_graph_export")
        return rowptr, col, perm, val

    def row_blocks(self, block_rows: int):
        """The forward structure cut into consecutive destination-row blocks of ``block_rows`` rows, each a shard
        structure of its own over the SAME source table: ``[(lo, hi, BucketedGraph), ...]``, built once (cached
        per block size) from the bucketed arrays - a block's edges are already in (row, relation, original
        column) order, so its stable re-bucketing keeps every segment's summation order.  The no-grad encoder
        walks these blocks (gather a block, transform it, next) so that the aggregate exists one cache-sized
        block at a time instead of as an ``[N, R * d]`` tensor in HBM."""
        if self.bipartite:
            raise ValueError("row blocks are cut from a whole graph")
        block_rows = max(32, (int(block_rows) + 31) // 32 * 32)
        cache = self.__dict__.setdefault("_row_blocks", {})
        if block_rows not in cache:
            n, r = self.num_nodes, self.num_relations
            rowptr, col, _, _ = self.arrays(False)
            rp = rowptr.long()
            blocks = []
            for lo in range(0, n, block_rows):
                hi = min(n, lo + block_rows)
                e0, e1 = int(rp[lo * r]), int(rp[hi * r])
                counts = rp[lo * r + 1: hi * r + 1] - rp[lo * r: hi * r]
                seg = torch.repeat_interleave(torch.arange((hi - lo) * r, device=self.device), counts)
                shard = BucketedGraph.from_shard(seg // r, col[e0:e1].long(), seg % r, hi - lo, n, r)
                blocks.append((lo, hi, shard))
            cache[block_rows] = blocks
        return cache[block_rows]

    def fused_plan(self, inline_limit: int = 16, transposed: bool = False) -> "FusedPlan":
        """What the one-kernel layer (``layer_fwd_fused`` / ``layer_bwd_input_fused``) needs beside this structure:
        which (node, relation) segments it walks itself (at most ``inline_limit`` <= 64 edges) and, for the longer
        ones, a gather structure of their own (one segment per long segment, one relation, key = row of the
        pre-aggregated ``hub`` table; weighted for the transposed direction) - built once per limit and direction
        from the bucketed arrays; a long segment keeps its edge order, run / pack cuts and hub reduce, so its
        aggregate has the bits the whole-graph gather gives it."""
        if self.weighted_shard:
            raise ValueError("the fused layer covers whole graphs and mean shards only")
        if transposed and self.bipartite:
            raise ValueError("a shard structure has one direction only (transposed=False)")
        limit = int(inline_limit)
        if not 1 <= limit <= 64:
            raise ValueError("inline_limit must be in [1, 64] (a longer walk is cut into runs by the gather)")
        cache = self.__dict__.setdefault("_fused_plans", {})
        key = (limit, bool(transposed))
        if key not in cache:
            rowptr, col, _, val = self.arrays(transposed)
            weight = val if transposed else None
            rp = rowptr.long()
            lens = rp[1:] - rp[:-1]
            long_seg = lens > limit
            hubs = int(long_seg.sum())
            if hubs == 0:
                cache[key] = FusedPlan(limit, rowptr, col, weight, None, 0, 0)
            else:
                hub_row = torch.cumsum(long_seg.long(), 0) - 1                  # of a long segment
                seg_of_edge = torch.repeat_interleave(torch.arange(lens.numel(), device=self.device), lens)
                in_hub = long_seg[seg_of_edge]
                hub = BucketedGraph.from_shard(hub_row[seg_of_edge[in_hub]], col[in_hub].long(),
                                               torch.zeros(int(in_hub.sum()), dtype=torch.int64, device=self.device),
                                               hubs, self.num_other_nodes, 1,
                                               edge_weight=weight[in_hub].contiguous() if transposed else None)
                # the walked CSR: short segments as they are, a long one as the single entry -(hub row + 1)
                first = torch.zeros_like(in_hub)
                first[rp[:-1][long_seg]] = True
                keep = ~in_hub | first
                new_col = torch.where(in_hub, -(hub_row[seg_of_edge] + 1), col.long())[keep].int()
                new_w = torch.where(in_hub, torch.ones_like(weight), weight)[keep].contiguous() if transposed else None
                new_lens = torch.where(long_seg, torch.ones_like(lens), lens)
                new_rowptr = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=self.device)
                new_rowptr[1:] = torch.cumsum(new_lens, 0)
                cache[key] = FusedPlan(limit, new_rowptr.int(), new_col.contiguous(), new_w, hub, hubs,
                                       int(in_hub.sum()))
        return cache[key]

    def merged_transposed(self) -> Optional["BucketedGraph"]:
        """The out-edges of every node across ALL relations as one weighted gather structure over
        a table of ``N * (R + 1)`` rows: edge (j -> i, r) reads row ``i * (R + 1) + r`` with weight
        ``1 / cnt[i, r]``, and every node j additionally reads its own row ``j * (R + 1) + R`` with
        weight 1.  With ``T = g @ [W_0^T | ... | W_{R-1}^T | root^T]`` viewed ``[N * (R + 1), d_in]``
        the layer's input gradient is ``aggregate(merged, T)`` - transform first, then gather
        ``d_in``-wide rows instead of ``d_out``-wide ones.  Built lazily from the transposed
        structure (already in (source, relation) order), once; None if it would not fit int32."""
        if self.bipartite:
            raise ValueError("a shard structure has no merged form")
        if getattr(self, "_merged", None) is None:
            n, r, e = self.num_nodes, self.num_relations, self.num_edges
            if n * (r + 1) >= 2 ** 31 - 1 or e + n >= 2 ** 31 - 1:
                return None
            rowptr_t, col_t, _, w_t = self.arrays(True)
            counts = (rowptr_t[1:] - rowptr_t[:-1]).long()
            seg = torch.repeat_interleave(torch.arange(n * r, device=self.device), counts)   # = src * R + rel
            nodes = torch.arange(n, device=self.device)
            key = torch.cat([seg // r, nodes])
            other = torch.cat([col_t.long() * (r + 1) + seg % r, nodes * (r + 1) + r])
            weight = torch.cat([w_t, torch.ones(n, device=self.device)])
            self._merged = BucketedGraph.from_shard(key, other, torch.zeros_like(key), n, n * (r + 1), 1, weight)
        return self._merged

    def destroy(self) -> None:
        merged = getattr(self, "_merged", None)
        if merged is not None:
            merged.destroy()
            self._merged = None
        for blocks in self.__dict__.pop("_row_blocks", {}).values():
            for _, _, shard in blocks:
                shard.destroy()
        for plan in self.__dict__.pop("_fused_plans", {}).values():
            if plan.hub is not None:
                plan.hub.destroy()
        if self._handle is not None:
            try:
                _L().rgcn_graph_destroy(self._handle)
            finally:
                self._handle = None

    def __del__(self):  # pragma: no cover - interpreter teardown order
        try:
            self.destroy()
        except Exception:
            pass


_GRAPH_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()
_GRAPH_CACHE_SIZE = 8


def _sidecar_matches(g: BucketedGraph, edge_index: torch.Tensor, edge_type: torch.Tensor) -> bool:
    """Does a loaded structure describe exactly these columns?  Bucketed edge k is original
    column perm[k] with source col[k], and (stable sort) perm ascends inside a segment, so the
    check is: perm is a permutation, sources match, and every edge sits in the segment of its
    (destination, relation)."""
    rowptr, col, perm, _ = g.arrays(False)
    e = g.num_edges
    if e == 0:
        return True
    if int(torch.bincount(perm, minlength=e).max()) != 1:
        return False
    seg = edge_index[1].index_select(0, perm) * g.num_relations + edge_type.index_select(0, perm)
    k = torch.arange(e, device=perm.device)
    lo, hi = rowptr[:-1].long().index_select(0, seg), rowptr[1:].long().index_select(0, seg)
    ok = (edge_index[0].index_select(0, perm) == col.long()).all() & ((k >= lo) & (k < hi)).all()
    ok &= (perm[1:] > perm[:-1])[seg[1:] == seg[:-1]].all()
    return bool(ok)


def bucket(edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int,
           num_relations: int, sidecar=None) -> BucketedGraph:
    """Cached bucketing.  The reference's graphs are constant for a whole run
    (``src/train.py:130-135`` holds three), so each is sorted once; the key follows the
    tensors' storage address and version counter, and the cache pins the tensors so an
    address cannot be recycled while its entry lives.

    ``sidecar``: path of the persisted structure next to the graph's ``.pt`` file.  If it
    exists and describes exactly these columns it is imported instead of sorting; otherwise
    the graph is bucketed and the file (re)written."""
    key = (edge_index.data_ptr(), edge_type.data_ptr(), edge_index._version, edge_type._version,
           tuple(edge_index.shape), int(num_nodes), int(num_relations), str(edge_index.device))
    hit = _GRAPH_CACHE.get(key)
    if hit is not None:
        _GRAPH_CACHE.move_to_end(key)
        return hit[0]
    g = None
    if sidecar is not None and os.path.exists(sidecar):
        _need_gpu("edge_index", edge_index, torch.int64)
        _need_gpu("edge_type", edge_type, torch.int64)
        try:
            cand = BucketedGraph.load(sidecar, edge_index.device)
            if ((cand.num_edges, cand.num_nodes, cand.num_relations)
                    == (edge_index.size(1), int(num_nodes), int(num_relations))
                    and _sidecar_matches(cand, edge_index, edge_type)):
                g = cand
        except (ValueError, IndexError, KeyError, RuntimeError, EOFError, pickle.UnpicklingError):
            g = None                                   # stale or damaged file: rebuild below
    if g is None:
        g = BucketedGraph(edge_index, edge_type, num_nodes, num_relations)
        if sidecar is not None:
            g.save(sidecar)
    _GRAPH_CACHE[key] = (g, edge_index, edge_type)
    while len(_GRAPH_CACHE) > _GRAPH_CACHE_SIZE:
        _GRAPH_CACHE.popitem(last=False)
    return g


def clear_graph_cache() -> None:
    _GRAPH_CACHE.clear()


# ----------------------------------------------------------------------------------
# aggregate (rows A3 + A4 / their autograd)
# ----------------------------------------------------------------------------------
# bench.py sets this to a list to collect (weighted, d, edges, segments, table rows, start_event, end_event) per
# level-0 gather launch; None (the default) means one C call per aggregate, no events.
GATHER_EVENTS = None
# likewise for the dense transforms: (kind, M, K, N, precision, start_event, end_event) per call - the
# bracket covers everything the call launches (split precision: operand scan, weight split, GEMM)
GEMM_EVENTS = None


class _GemmBracket:
    """HIP events around one transform call on the launch stream, only while bench.py collects them"""

    def __new__(cls, *info):
        return _NO_GUARD if GEMM_EVENTS is None else super().__new__(cls)   # nothing to construct outside bench.py

    def __init__(self, kind: str, m: int, k: int, n: int, precision: str):
        self.info = (kind, m, k, n, precision)

    def __enter__(self):
        if GEMM_EVENTS is not None:
            self.beg, self.end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.beg.record()
        return self

    def __exit__(self, *exc):
        if GEMM_EVENTS is not None and exc[0] is None:
            self.end.record()
            GEMM_EVENTS.append(self.info + (self.beg, self.end))
        return False


class PendingParamGrads:
    """Parameter gradients whose slab reduction has not been launched yet
    (``transform_bwd_params(..., defer=True)``).  Hand it to the next ``aggregate`` of the same
    backward (``tail=``): the 5 us reduction then rides in that gather launch as extra
    workgroups instead of sitting between two launch boundaries; or call ``finish()``.
    ``grads`` = (grad_weight, grad_root | None, grad_bias | None), valid once either happened."""

    def __init__(self, grads, job, workspace):
        self.grads, self.job, self._workspace, self.done = grads, job, workspace, False

    def finish(self) -> None:
        if not self.done:
            dev = self.grads[0].device
            with _on(dev):
                rc = _L().rgcn_slab_reduce(ctypes.byref(self.job), _stream())
            _lib.check(rc, "rgcn_slab_reduce")
            self._launched()

    def _launched(self) -> None:
        self.done, self._workspace = True, None        # stream-ordered allocator: safe to release after the launch


def aggregate(graph: BucketedGraph, x: torch.Tensor, transposed: bool = False,
              tail: Optional[PendingParamGrads] = None, amax_out: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[N, R*d]``: per-(dst, rel) mean of source rows (``transposed=False``) or the
    1/cnt-weighted sum over out-edges per (src, rel) (``transposed=True``).  For a shard
    (``BucketedGraph.from_shard``) x holds the gathered rows of all ranks and the result has
    the rank's own rows.  ``x`` may be float16 (BASELINE configs[4]: fp16 feature table, half
    the bytes per gathered row); sums and the result are fp32 either way.  ``amax_out`` (a ZEROED amax
    buffer, fp32 table only): receives ``max |result|``, the scale the split-precision
    transforms need for this operand."""
    half_in = isinstance(x, torch.Tensor) and x.dtype == torch.float16
    _need_gpu("x", x, torch.float16 if half_in else torch.float32)
    if graph.bipartite and transposed:
        raise ValueError("a shard structure has one direction only (transposed=False)")
    if x.dim() != 2 or x.size(0) != graph.num_other_nodes:
        raise ValueError(f"x must be [{graph.num_other_nodes}, d], got {tuple(x.shape)}")
    if x.device != graph.device:
        raise RuntimeError("x and the bucketed graph are on different devices")
    d = x.size(1)
    if d % (8 if half_in else 4):
        raise ValueError(f"feature dim {d} must be a multiple of {8 if half_in else 4}")
    _check_amax("amax_out", amax_out, x.device)
    if amax_out is not None and half_in:
        raise ValueError("amax_out goes with the fp32 gather (the fp16 path's transform needs no scale)")
    lib = _L()
    if out is not None:
        _need_gpu("out", out, torch.float32)
        if tuple(out.shape) != (graph.num_nodes, graph.num_relations * d) or out.device != x.device:
            raise ValueError(f"out must be [{graph.num_nodes}, {graph.num_relations * d}] on x's device")
    with _on(x.device):
        if out is None:
            out = _empty(graph.num_nodes, graph.num_relations * d, dtype=torch.float32, device=x.device)
        nbytes = graph.workspace_bytes(transposed, d)
        ws = _workspace(nbytes, x.device)
        if tail is not None and not tail.done and (half_in or GATHER_EVENTS is not None):
            tail.finish()                       # no ride in these modes: launch the reduction by itself
        if half_in:
            rc = lib.rgcn_aggregate_f16(graph.handle, int(transposed), _ptr(x), d, _ptr(out), _ptr(ws), nbytes,
                                        _stream())
        elif GATHER_EVENTS is None and amax_out is not None:
            job = tail.job if (tail is not None and not tail.done) else None
            rc = lib.rgcn_aggregate_amax(graph.handle, int(transposed), _ptr(x), d, _ptr(out), _ptr(ws), nbytes,
                                         ctypes.byref(job) if job is not None else None, _ptr(amax_out), _stream())
            if rc == 0 and job is not None:
                tail._launched()
        elif GATHER_EVENTS is None and tail is not None and not tail.done:
            rc = lib.rgcn_aggregate_and_reduce(graph.handle, int(transposed), _ptr(x), d, _ptr(out), _ptr(ws), nbytes,
                                               ctypes.byref(tail.job), _stream())
            if rc == 0:
                tail._launched()
        elif GATHER_EVENTS is None:
            rc = lib.rgcn_aggregate(graph.handle, int(transposed), _ptr(x), d, _ptr(out), _ptr(ws), nbytes,
                                    _stream())
        else:
            # measurement mode (bench.py): same launches, level 0 (the gather kernel proper)
            # bracketed by HIP events on the stream it is launched on
            rc = 0
            for level in range(graph.num_levels(transposed)):
                if level == 0:
                    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    beg.record()
                rc = rc or lib.rgcn_aggregate_level(graph.handle, int(transposed), level, _ptr(x), d,
                                                    _ptr(out), _ptr(ws), nbytes, _ptr(amax_out), _stream())
                if level == 0:
                    end.record()
                    weighted = bool(transposed) or (graph.bipartite and graph.weighted_shard)
                    GATHER_EVENTS.append((weighted, d, graph.num_edges, graph.num_nodes * graph.num_relations,
                                          graph.num_other_nodes, beg, end))
    _lib.check(rc, "rgcn_aggregate")
    return out


def fused_bwd_supported(num_relations: int, d_in: int, d_out: int) -> bool:
    """does the one-kernel input gradient cover this layer shape (in split precision)?"""
    return GEMM_PRECISION != "fp32" and bool(_query("rgcn_layer_bwd_input_fused_supported", int(num_relations), int(d_in), int(d_out)))


FUSED_EVENTS = None      # bench / probes: list that receives (kind, rows, relations, inline edges, pre-aggregated rows,
                         # gathered width, output width, begin, end) per fused launch


def layer_fwd_fused(graph: BucketedGraph, x: torch.Tensor, packed: SplitWeights, bias: Optional[torch.Tensor],
                    relu: bool, amax: torch.Tensor, amax_out: Optional[torch.Tensor] = None,
                    inline_limit: int = 16, out: Optional[torch.Tensor] = None,
                    agg_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[mean-aggregate(x) | x] @ [W ; root] + bias`` (+ ReLU) in ONE kernel whose A operand is gathered into
    LDS (``rgcn_layer_fwd_fused``): no ``[N, R * d_in]`` aggregate in HBM.  Segments longer than ``inline_limit``
    edges are pre-aggregated by the ordinary gather over ``graph.fused_plan(inline_limit).hub``.  Bit-identical
    to ``aggregate`` -> ``transform_fwd(precision="split")``.  ``amax``: amax buffer of ``x``.  ``agg_out``
    (``[N, R * d_in]``, optional): also receives the aggregate, for a backward that needs it."""
    _need_gpu("x", x, torch.float32)
    if graph.weighted_shard:
        raise ValueError("the fused layer covers mean structures only")
    if x.dim() != 2 or x.size(0) != graph.num_other_nodes or x.device != graph.device:
        raise ValueError(f"x must be [{graph.num_other_nodes}, d] on the graph's device, got {tuple(x.shape)}")
    r, d_in, d_out = packed.shape
    if r != graph.num_relations or d_in != x.size(1):
        raise ValueError(f"packed weights are [{r}, {d_in}, {d_out}]; graph has {graph.num_relations} relations, x width {x.size(1)}")
    if bias is not None:
        _need_gpu("bias", bias, torch.float32)
        if tuple(bias.shape) != (d_out,):
            raise ValueError(f"bias must be [{d_out}]")
    _check_amax("amax", amax, x.device)
    if amax is None:
        raise ValueError("amax (the amax buffer of x) is required")
    _check_amax("amax_out", amax_out, x.device)
    if agg_out is not None:
        _need_gpu("agg_out", agg_out, torch.float32)
        if tuple(agg_out.shape) != (graph.num_nodes, r * d_in) or agg_out.device != x.device or not agg_out.is_contiguous():
            raise ValueError(f"agg_out must be a contiguous [{graph.num_nodes}, {r * d_in}] on x's device")
    plan = graph.fused_plan(min(int(inline_limit), d_in // 4))    # one id window of d_in / 4 lanes per segment
    hub_agg = aggregate(plan.hub, x) if plan.hub is not None else None
    tile_mask = graph.tile_mask_ptr(False) if graph.num_relations <= 32 else None
    lib = _L()
    with _on(x.device):
        if out is None:
            out = _empty(graph.num_nodes, d_out, dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (graph.num_nodes, d_out) or out.dtype != torch.float32 or out.device != x.device:
            raise ValueError(f"out must be float32 [{graph.num_nodes}, {d_out}] on x's device")
        if FUSED_EVENTS is not None:
            beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            beg.record()
        rc = lib.rgcn_layer_fwd_fused(_ptr(plan.rowptr), _ptr(plan.col), tile_mask, graph.num_nodes, r, _ptr(hub_agg),
                                      _ptr(x), _ptr(packed.buf), int(packed.has_root), _ptr(bias),
                                      int(bool(relu)), d_in, d_out, _ptr(amax), _ptr(out), _ptr(amax_out), _ptr(agg_out),
                                      _stream())
        if FUSED_EVENTS is not None:
            end.record()
            FUSED_EVENTS.append(("fwd+store" if agg_out is not None else "fwd", graph.num_nodes, graph.num_relations,
                                 graph.num_edges - plan.hub_edges, plan.hub_rows, d_in, d_out, beg, end))
    _lib.check(rc, "rgcn_layer_fwd_fused")
    return out


def layer_bwd_input_fused(graph: BucketedGraph, g: torch.Tensor, packed: SplitWeights,
                          relu_mask: Optional[torch.Tensor], amax: torch.Tensor,
                          amax_out: Optional[torch.Tensor] = None, inline_limit: int = 16,
                          tail: Optional["PendingParamGrads"] = None, out_scale: float = 1.0) -> torch.Tensor:
    """``grad_x = out_scale * [transposed-aggregate(g) | g] @ [W_r^T ; root^T]`` (``* (relu_mask > 0)``) in ONE kernel
    (``rgcn_layer_bwd_input_fused``): the weighted sums over out-edges are formed in LDS, no ``[N, R * d_out]``
    tensor in HBM.  Bit-identical to ``aggregate(transposed=True)`` -> ``transform_bwd_input(precision="split")``.
    ``amax``: amax buffer of ``g``; ``tail``: a pending parameter-gradient reduction that rides in the gather of
    the long segments (or is launched by itself if there is none)."""
    _need_gpu("g", g, torch.float32)
    if graph.bipartite:
        raise ValueError("the fused input gradient needs the transposed structure of a whole graph")
    r, d_in, d_out = packed.shape
    if g.dim() != 2 or tuple(g.shape) != (graph.num_nodes, d_out) or g.device != graph.device or r != graph.num_relations:
        raise ValueError(f"g must be [{graph.num_nodes}, {d_out}] on the graph's device with {r} relations")
    if relu_mask is not None:
        _need_gpu("relu_mask", relu_mask, torch.float32)
        if tuple(relu_mask.shape) != (graph.num_nodes, d_in) or not relu_mask.is_contiguous():
            raise ValueError(f"relu_mask must be a contiguous [{graph.num_nodes}, {d_in}]")
    _check_amax("amax", amax, g.device)
    if amax is None:
        raise ValueError("amax (the amax buffer of g) is required")
    _check_amax("amax_out", amax_out, g.device)
    plan = graph.fused_plan(min(int(inline_limit), d_out // 4), transposed=True)
    hub_agg = aggregate(plan.hub, g, tail=tail) if plan.hub is not None else None   # a pending slab reduction rides along
    if tail is not None and not tail.done:
        tail.finish()
    tile_mask = graph.tile_mask_ptr(True) if graph.num_relations <= 32 else None
    lib = _L()
    with _on(g.device):
        gx = _empty(graph.num_nodes, d_in, dtype=torch.float32, device=g.device)
        if FUSED_EVENTS is not None:
            beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            beg.record()
        rc = lib.rgcn_layer_bwd_input_fused(_ptr(plan.rowptr), _ptr(plan.col), _ptr(plan.weight), tile_mask,
                                            graph.num_nodes, r, _ptr(hub_agg), _ptr(g), _ptr(packed.buf),
                                            int(packed.has_root), _ptr(relu_mask), d_in, d_out, _ptr(amax),
                                            float(graph.weight_bound(True)), _ptr(gx), _ptr(amax_out), _stream(),
                                            float(out_scale))
        if FUSED_EVENTS is not None:
            end.record()
            FUSED_EVENTS.append(("bwd_input+mask" if relu_mask is not None else "bwd_input", graph.num_nodes,
                                 graph.num_relations, graph.num_edges - plan.hub_edges, plan.hub_rows, d_out, d_in, beg, end))
    _lib.check(rc, "rgcn_layer_bwd_input_fused")
    return gx


class DeferredHubs:
    """What ``aggregate_deferred`` leaves for the transform that consumes its result: the structure and direction
    (their level-1 items by row tile) and the partial rows (kept alive here until that transform has been launched)."""

    def __init__(self, graph: BucketedGraph, transposed: bool, partial: torch.Tensor):
        self.graph, self.transposed, self.partial = graph, bool(transposed), partial


def aggregate_deferred(graph: BucketedGraph, x: torch.Tensor, transposed: bool = False,
                       tail: Optional[PendingParamGrads] = None):
    """``aggregate`` WITHOUT the hub-tail launch, for a result that goes straight into ``transform_fwd`` /
    ``transform_bwd_input`` in split precision: ``(agg, hubs)`` - hand ``hubs`` to that call, which sums the partial
    rows of the long segments tile by tile itself (same order, same bits) and completes ``agg`` in passing.
    ``hubs`` is None (and ``agg`` complete) where nothing can be deferred: no long segment, a structure with more
    than one reduce level, a width other than 64 / 128 / 256, an fp16 table, measurement mode."""
    lib = _L()
    d = x.size(1) if x.dim() == 2 else 0
    deferrable = (x.dtype == torch.float32 and GATHER_EVENTS is None and graph.num_levels(transposed) == 2
                  and graph.deferrable(transposed, d))
    if not deferrable:
        return aggregate(graph, x, transposed, tail=tail), None
    _need_gpu("x", x, torch.float32)
    if graph.bipartite and transposed:
        raise ValueError("a shard structure has one direction only (transposed=False)")
    if x.size(0) != graph.num_other_nodes or x.device != graph.device:
        raise ValueError(f"x must be [{graph.num_other_nodes}, d] on the graph's device, got {tuple(x.shape)}")
    with _on(x.device):
        out = _empty(graph.num_nodes, graph.num_relations * d, dtype=torch.float32, device=x.device)
        nbytes = graph.workspace_bytes(transposed, d)
        ws = _workspace(nbytes, x.device)
        job = tail.job if (tail is not None and not tail.done) else None
        rc = lib.rgcn_aggregate_deferred(graph.handle, int(transposed), _ptr(x), d, _ptr(out), _ptr(ws), nbytes,
                                         ctypes.byref(job) if job is not None else None, _stream())
        if rc == 0 and job is not None:
            tail._launched()
    _lib.check(rc, "rgcn_aggregate_deferred")
    return out, DeferredHubs(graph, transposed, ws)


# 1: the pass's first launch rides in conv1's gather (rgcn_aggregate_prep).  OFF by default: measured on the MI355X the
# step gets 5 us SLOWER (0.286 against 0.281 ms, profiles/r03_prep_rides.txt) - in 256-thread workgroups the riders' scan
# of the weights takes 8 rounds of loads instead of 2 and outlasts the gather it was meant to hide behind.
def _hub_args(hubs: Optional[DeferredHubs], n: int, r: int):
    if hubs is None:
        return None, 0, None
    if hubs.graph.num_nodes != n or hubs.graph.num_relations != r:
        raise ValueError("hubs do not belong to this aggregate")
    return hubs.graph.handle, int(hubs.transposed), _ptr(hubs.partial)


# ----------------------------------------------------------------------------------
# transform (row A6 / its autograd)
# ----------------------------------------------------------------------------------
def _check_layer(agg, x, weight, root, bias):
    _need_gpu("x", x, torch.float32)
    _need_gpu("agg", agg, torch.float32)
    _need_gpu("weight", weight, torch.float32)
    if weight.dim() != 3:
        raise ValueError("weight must be [R, d_in, d_out]")
    r, d_in, d_out = weight.shape
    n = x.size(0)
    if x.dim() != 2 or x.size(1) != d_in:
        raise ValueError(f"x must be [N, {d_in}], got {tuple(x.shape)}")
    if tuple(agg.shape) != (n, r * d_in):
        raise ValueError(f"agg must be [{n}, {r * d_in}], got {tuple(agg.shape)}")
    if root is not None:
        _need_gpu("root", root, torch.float32)
        if tuple(root.shape) != (d_in, d_out):
            raise ValueError(f"root must be [{d_in}, {d_out}]")
    if bias is not None:
        _need_gpu("bias", bias, torch.float32)
        if tuple(bias.shape) != (d_out,):
            raise ValueError(f"bias must be [{d_out}]")
    if d_in % 4 or d_out % 4:
        raise ValueError("in/out channels must be multiples of 4")
    return n, r, d_in, d_out


def _mask_for(graph: Optional[BucketedGraph], transposed: bool, n: int, r: int) -> Optional[int]:
    """relation-occupancy mask of the structure the aggregate came from (None = dense)"""
    if graph is None:
        return None
    if graph.num_nodes != n or graph.num_relations != r:
        raise ValueError("graph does not match the operand shapes")
    return graph.tile_mask_ptr(transposed and not graph.bipartite)


def transform_fwd(agg, x, weight, root=None, bias=None, relu: bool = False,
                  graph: Optional[BucketedGraph] = None, half: bool = False, amax=None,
                  amax_out: Optional[torch.Tensor] = None, precision: Optional[str] = None,
                  packed: Optional[SplitWeights] = None, amax_mul: float = 1.0,
                  out: Optional[torch.Tensor] = None, hubs: Optional[DeferredHubs] = None) -> torch.Tensor:
    """``sum_r agg[:, r] @ weight[r] + x @ root + bias`` as one fp32-MFMA GEMM; ``relu``
    fuses the activation that follows conv1 (``rgcn.py:124``) into the epilogue.  ``graph``
    (the structure ``agg`` was aggregated over) lets the kernel skip the k-tiles of relations
    that a whole 64-row tile does not have - exact zeros in ``agg``.

    ``half=True`` (BASELINE configs[4]): operands rounded to fp16, fp32 accumulate, on the fp16
    matrix cores (``rgcn_transform_fwd_f16``); widths the kernel does not take (d_in % 32) run the
    fp32 GEMM instead - more precise, never less.

    Split precision (``GEMM_PRECISION``): ``amax = (agg_amax, x_amax)`` are amax buffers
    (``absmax``, a previous transform's ``amax_out``, ``aggregate(amax_out=)``); ``agg`` is scaled by
    the bound ``amax_mul * value(agg_amax)`` - pass the buffer of the table ``agg`` was gathered from
    and the structure's ``weight_bound`` (a mean of rows cannot exceed the table's maximum: 1) - and
    ``x`` by its own maximum; a missing buffer is scanned here (one more pass over that operand).
    ``amax_out`` (a ZEROED amax buffer): receives ``max |out|``."""
    n, r, d_in, d_out = _check_layer(agg, x, weight, root, bias)
    lib = _L()
    _check_amax("amax_out", amax_out, x.device)
    if out is not None:
        _need_gpu("out", out, torch.float32)
        if tuple(out.shape) != (n, d_out) or out.device != x.device:
            raise ValueError(f"out must be [{n}, {d_out}] on x's device")
    given_out = out
    split = 0 if half else _use_split(precision, d_in, 32)
    if split and n > 0:
        a1, a2 = amax if amax is not None else (None, None)
        _check_amax("agg_amax", a1, x.device)
        _check_amax("x_amax", a2, x.device)
        if packed is not None and not packed.matches(weight, root):
            raise ValueError("packed does not belong to these weights")
        with _on(x.device):
            out = given_out if given_out is not None else _empty(n, d_out, dtype=torch.float32, device=x.device)
            nbytes = _query("rgcn_transform_split_workspace_bytes", r, d_in, d_out)
            ws = _workspace(nbytes, x.device)
            with _GemmBracket("fwd", n, (r + (root is not None)) * d_in, d_out, "split" if split == 1 else "half"):
                rc = lib.rgcn_transform_fwd_split(_ptr(agg), _ptr(x), _ptr(weight), _ptr(root),
                                                  _ptr(packed.buf) if packed is not None else None, _ptr(bias),
                                                  int(relu), _mask_for(graph, False, n, r), n, r, d_in, d_out,
                                                  _ptr(a1), float(amax_mul), _ptr(a2), int(split == 2), _ptr(out),
                                                  _ptr(amax_out), _ptr(ws), nbytes, _stream(), *_hub_args(hubs, n, r))
        _lib.check(rc, "rgcn_transform_fwd_split")
        return out
    if hubs is not None:
        raise ValueError("deferred hub tails need the split-precision transform (finish them with aggregate instead)")
    if half and d_in % 32 == 0:
        with _on(x.device):
            out = given_out if given_out is not None else _empty(n, d_out, dtype=torch.float32, device=x.device)
            nbytes = lib.rgcn_transform_fwd_f16_workspace_bytes(r, d_in, d_out)
            ws = _workspace(nbytes, x.device)
            with _GemmBracket("fwd", n, (r + (root is not None)) * d_in, d_out, "f16"):
                rc = lib.rgcn_transform_fwd_f16(_ptr(agg), _ptr(x), _ptr(weight), _ptr(root), _ptr(bias), int(relu),
                                                _mask_for(graph, False, n, r), n, r, d_in, d_out, _ptr(out), _ptr(ws),
                                                nbytes, _stream())
        _lib.check(rc, "rgcn_transform_fwd_f16")
        if amax_out is not None:
            absmax(out, amax_out)
        return out
    with _on(x.device):
        out = given_out if given_out is not None else _empty(n, d_out, dtype=torch.float32, device=x.device)
        with _GemmBracket("fwd", n, (r + (root is not None)) * d_in, d_out, "fp32"):
            rc = lib.rgcn_transform_fwd(_ptr(agg), _ptr(x), _ptr(weight), _ptr(root), _ptr(bias), int(relu),
                                        _mask_for(graph, False, n, r), n, r, d_in, d_out, _ptr(out), _stream())
    _lib.check(rc, "rgcn_transform_fwd")
    if amax_out is not None:
        absmax(out, amax_out)
    return out


def transform_bwd_input(gagg, g, weight, root=None, relu_mask=None,
                        graph: Optional[BucketedGraph] = None, amax=None, amax_out: Optional[torch.Tensor] = None,
                        precision: Optional[str] = None, packed: Optional[SplitWeights] = None,
                        amax_mul: float = 1.0, hubs: Optional[DeferredHubs] = None,
                        out_scale: float = 1.0) -> torch.Tensor:
    """``grad_x = out_scale * (sum_r gagg[:, r] @ weight[r]^T + g @ root^T)``; with ``relu_mask`` (the
    layer's input, when that input is the output of a fused-ReLU layer) the result is
    additionally multiplied by ``relu_mask > 0``.  ``out_scale`` (the ``1 / (1 - p)`` of a dropout whose
    backward rides in this call) is applied in the split kernels' epilogue; the fp32 kernels get scaled
    weights instead."""
    _need_gpu("g", g, torch.float32)
    _need_gpu("gagg", gagg, torch.float32)
    _need_gpu("weight", weight, torch.float32)
    r, d_in, d_out = weight.shape
    n = g.size(0)
    if tuple(g.shape) != (n, d_out) or tuple(gagg.shape) != (n, r * d_out):
        raise ValueError("g must be [N, d_out] and gagg [N, R*d_out]")
    if root is not None:
        _need_gpu("root", root, torch.float32)
    if relu_mask is not None:
        _need_gpu("relu_mask", relu_mask, torch.float32)
        if tuple(relu_mask.shape) != (n, d_in):
            raise ValueError(f"relu_mask must be [{n}, {d_in}]")
    lib = _L()
    _check_amax("amax_out", amax_out, g.device)
    if d_in % 4 or d_out % 4:
        raise ValueError("in/out channels must be multiples of 4")
    split = _use_split(precision, d_out, 32)
    if split and n > 0:
        a1, a2 = amax if amax is not None else (None, None)
        _check_amax("gagg_amax", a1, g.device)
        _check_amax("g_amax", a2, g.device)
        if packed is not None and not packed.matches(weight, root):
            raise ValueError("packed does not belong to these weights")
        with _on(g.device):
            gx = _empty(n, d_in, dtype=torch.float32, device=g.device)
            nbytes = _query("rgcn_transform_split_workspace_bytes", r, d_in, d_out)
            ws = _workspace(nbytes, g.device)
            with _GemmBracket("bwd_input", n, (r + (root is not None)) * d_out, d_in, "split" if split == 1 else "half"):
                rc = lib.rgcn_transform_bwd_input_split(_ptr(gagg), _ptr(g), _ptr(weight), _ptr(root),
                                                        _ptr(packed.buf) if packed is not None else None,
                                                        _ptr(relu_mask), _mask_for(graph, True, n, r), n, r, d_in,
                                                        d_out, _ptr(a1), float(amax_mul), _ptr(a2), int(split == 2),
                                                        _ptr(gx), _ptr(amax_out), _ptr(ws), nbytes, _stream(),
                                                        *_hub_args(hubs, n, r), float(out_scale))
        _lib.check(rc, "rgcn_transform_bwd_input_split")
        return gx
    if hubs is not None:
        raise ValueError("deferred hub tails need the split-precision transform (finish them with aggregate instead)")
    if out_scale != 1.0:                                   # the fp32 kernels have no output factor: scale the (small) weights
        weight = weight * out_scale
        root = root * out_scale if root is not None else None
    with _on(g.device):
        gx = _empty(n, d_in, dtype=torch.float32, device=g.device)
        with _GemmBracket("bwd_input", n, (r + (root is not None)) * d_out, d_in, "fp32"):
            rc = lib.rgcn_transform_bwd_input(_ptr(gagg), _ptr(g), _ptr(weight), _ptr(root), _ptr(relu_mask),
                                              _mask_for(graph, True, n, r), n, r, d_in, d_out, _ptr(gx), _stream())
    _lib.check(rc, "rgcn_transform_bwd_input")
    if amax_out is not None:
        absmax(gx, amax_out)
    return gx


def chain_supported(weight: torch.Tensor, weight1: torch.Tensor) -> bool:
    """can conv2's input gradient (``weight`` [R, d_in, d_out]) carry conv1's transform-first product (``weight1``
    [R1, d_in1, d_in]) behind it in one launch (``transform_bwd_input_chain``)?  split precision, d_in == 128"""
    r, d_in, d_out = weight.shape
    r1, d_in1, d_out1 = weight1.shape
    return bool(GEMM_PRECISION == "split" and d_out1 == d_in and
                _query("rgcn_transform_bwd_input_chain_supported", r, d_in, d_out, r1, d_in1))


def transform_bwd_input_chain(gagg, g, weight, root, relu_mask, packed: SplitWeights, packed1: SplitWeights,
                              graph: Optional[BucketedGraph] = None, amax=None, amax_out: Optional[torch.Tensor] = None,
                              amax_mul: float = 1.0, hubs: Optional[DeferredHubs] = None, out_scale: float = 1.0):
    """``(gz, T)``: ``gz = transform_bwd_input(gagg, g, weight, root, relu_mask, ...)`` - the same bits - and
    ``T = transform_first(gz, packed1)``, conv1's transform-first product (``[N, (R1 + 1) * d_in1]``), computed by the
    SAME launch: every workgroup keeps its 64-row tile of ``gz`` as fp16 hi / lo fragments and multiplies it by conv1's
    weights right away (``rgcn_transform_bwd_input_chain_split``; 47.6-49.5 -> 43.2 us at C2)."""
    _need_gpu("g", g, torch.float32)
    _need_gpu("gagg", gagg, torch.float32)
    _need_gpu("weight", weight, torch.float32)
    r, d_in, d_out = weight.shape
    r1, d_in1, _ = packed1.shape
    n = g.size(0)
    if tuple(g.shape) != (n, d_out) or tuple(gagg.shape) != (n, r * d_out):
        raise ValueError("g must be [N, d_out] and gagg [N, R*d_out]")
    if root is not None:
        _need_gpu("root", root, torch.float32)
    if relu_mask is not None:
        _need_gpu("relu_mask", relu_mask, torch.float32)
        if tuple(relu_mask.shape) != (n, d_in):
            raise ValueError(f"relu_mask must be [{n}, {d_in}]")
    if not packed.matches(weight, root) or packed1.shape[2] != d_in:
        raise ValueError("packed / packed1 do not belong to these layers")
    a1, a2 = amax if amax is not None else (None, None)
    _check_amax("gagg_amax", a1, g.device)
    _check_amax("g_amax", a2, g.device)
    _check_amax("amax_out", amax_out, g.device)
    if a1 is None or a2 is None:
        raise ValueError("the chained transform needs the operand maxima (amax=)")
    lib = _L()
    cols = (r1 + (1 if packed1.has_root else 0)) * d_in1
    with _on(g.device):
        gz = _empty(n, d_in, dtype=torch.float32, device=g.device)
        t = _empty(n, cols, dtype=torch.float32, device=g.device)
        nbytes = _query("rgcn_transform_split_workspace_bytes", r, d_in, d_out)
        ws = _workspace(nbytes, g.device)
        # (bench.py's bracket: K + cols reduction columns against d_in outputs = the flops and operand bytes of both products)
        with _GemmBracket("bwd_input_chain", n, (r + (root is not None)) * d_out + cols, d_in, "split"):
            rc = lib.rgcn_transform_bwd_input_chain_split(_ptr(gagg), _ptr(g), _ptr(weight), _ptr(root), _ptr(packed.buf),
                                                          _ptr(relu_mask), _mask_for(graph, True, n, r), n, r, d_in, d_out,
                                                          _ptr(a1), float(amax_mul), _ptr(a2), _ptr(gz), _ptr(amax_out), _ptr(ws),
                                                          nbytes, _stream(), *_hub_args(hubs, n, r), float(out_scale),
                                                          _ptr(packed1.buf), int(packed1.has_root), r1, d_in1, _ptr(t))
    _lib.check(rc, "rgcn_transform_bwd_input_chain_split")
    return gz, t


def transform_first(g: torch.Tensor, packed: SplitWeights, amax: Optional[torch.Tensor] = None,
                    precision: Optional[str] = None) -> torch.Tensor:
    """``T = g @ [W_0^T | ... | W_{R-1}^T | root^T]`` -> ``[N, (R + 1) * d_in]`` (``R * d_in`` without a root) from
    the split weights' natural-order image (``rgcn_transform_first_split``): the dense half of the transform-first
    input gradient, without concatenating or splitting the weights again.  ``amax``: amax buffer of ``g``."""
    _need_gpu("g", g, torch.float32)
    r, d_in, d_out = packed.shape
    if g.dim() != 2 or g.size(1) != d_out or not g.is_contiguous() or g.device != packed.buf.device:
        raise ValueError(f"g must be a contiguous [N, {d_out}] on the weights' device")
    split = _use_split(precision, d_out, 32)
    if not split:
        raise ValueError("transform_first runs in split precision only")
    _check_amax("amax", amax, g.device)
    lib = _L()
    n = g.size(0)
    with _on(g.device):
        t = _empty(n, (r + int(packed.has_root)) * d_in, dtype=torch.float32, device=g.device)
        ws = _workspace(2048, g.device)
        with _GemmBracket("bwd_input", n, d_out, t.size(1), "split" if split == 1 else "half"):
            rc = lib.rgcn_transform_first_split(_ptr(g), _ptr(packed.buf), int(packed.has_root), n, r, d_in, d_out,
                                                _ptr(amax), int(split == 2), _ptr(t), _ptr(ws), 2048, _stream())
    _lib.check(rc, "rgcn_transform_first_split")
    return t


def transform_bwd_params(agg, x, g, num_relations: int, want_root: bool = True, want_bias: bool = True,
                         graph: Optional[BucketedGraph] = None, defer: bool = False, amax=None,
                         precision: Optional[str] = None, amax_mul: float = 1.0):
    """``(grad_weight[R, d_in, d_out], grad_root | None, grad_bias | None)``; with ``defer=True`` a
    ``PendingParamGrads`` whose reduction the caller attaches to the next gather (or finishes).
    Split precision: ``amax = (agg_amax, x_amax, g_amax)``, any of them None (scanned here)."""
    _need_gpu("x", x, torch.float32)
    _need_gpu("agg", agg, torch.float32)
    _need_gpu("g", g, torch.float32)
    n, d_in = x.shape
    d_out = g.size(1)
    r = int(num_relations)
    if tuple(agg.shape) != (n, r * d_in) or g.size(0) != n:
        raise ValueError("agg must be [N, R*d_in] and g [N, d_out]")
    lib = _L()
    with _on(x.device):
        gw = _empty(r, d_in, d_out, dtype=torch.float32, device=x.device)
        groot = _empty(d_in, d_out, dtype=torch.float32, device=x.device) if want_root else None
        gbias = _empty(d_out, dtype=torch.float32, device=x.device) if want_bias else None
        split = _use_split(precision, d_in, 64)
        if split and n > 0:
            a1, a2, a3 = amax if amax is not None else (None, None, None)
            for nm, t in (("agg_amax", a1), ("x_amax", a2), ("g_amax", a3)):
                _check_amax(nm, t, x.device)
            nbytes = _query("rgcn_transform_bwd_params_split_workspace_bytes", n, r, d_in, d_out)
            ws = _workspace(nbytes, x.device)
            job = _lib.SlabJob()
            with _GemmBracket("bwd_params", (r + want_root) * d_in, n, d_out, "split" if split == 1 else "half"):
                rc = lib.rgcn_transform_bwd_params_split_begin(_ptr(agg), _ptr(x), _ptr(g),
                                                               _mask_for(graph, False, n, r), n, r, d_in, d_out,
                                                               _ptr(a1), float(amax_mul), _ptr(a2), _ptr(a3),
                                                               int(split == 2), _ptr(gw), _ptr(groot),
                                                               _ptr(gbias), _ptr(ws), nbytes, _stream(),
                                                               ctypes.byref(job))
            _lib.check(rc, "rgcn_transform_bwd_params_split_begin")
            pending = PendingParamGrads((gw, groot, gbias), job, ws)
            if defer:
                return pending
            pending.finish()
            return gw, groot, gbias
        nbytes = lib.rgcn_transform_bwd_params_workspace_bytes(n, r, d_in, d_out)
        ws = _workspace(nbytes, x.device)
        if defer:
            job = _lib.SlabJob()
            with _GemmBracket("bwd_params", (r + want_root) * d_in, n, d_out, "fp32"):
                rc = lib.rgcn_transform_bwd_params_begin(_ptr(agg), _ptr(x), _ptr(g), _mask_for(graph, False, n, r), n,
                                                         r, d_in, d_out, _ptr(gw), _ptr(groot), _ptr(gbias), _ptr(ws),
                                                         nbytes, _stream(), ctypes.byref(job))
            _lib.check(rc, "rgcn_transform_bwd_params_begin")
            return PendingParamGrads((gw, groot, gbias), job, ws)
        rc = lib.rgcn_transform_bwd_params(_ptr(agg), _ptr(x), _ptr(g), _mask_for(graph, False, n, r), n, r, d_in,
                                           d_out, _ptr(gw),
                                           _ptr(groot), _ptr(gbias), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "rgcn_transform_bwd_params")
    return gw, groot, gbias


# ----------------------------------------------------------------------------------
# DistMult head (rows C1 + C2)
# ----------------------------------------------------------------------------------
def _check_operand(name, mat, idx, batch):
    _need_gpu(name, mat, torch.float32)
    if mat.dim() != 2:
        raise ValueError(f"{name} must be 2-D")
    if idx is not None:
        _need_gpu(name + "_idx", idx, torch.int64)
        if idx.dim() != 1 or idx.size(0) != batch:
            raise ValueError(f"{name}_idx must be [{batch}]")
    elif mat.size(0) != batch:
        raise ValueError(f"{name} must have {batch} rows when no index is given")


def distmult_fwd(h, h_idx, t, t_idx, r, r_idx, batch: int) -> torch.Tensor:
    d = h.size(1)
    for name, m, i in (("head", h, h_idx), ("tail", t, t_idx), ("rel", r, r_idx)):
        _check_operand(name, m, i, batch)
        if m.size(1) != d:
            raise ValueError("head / tail / relation embedding dims differ")
    if d % 4:
        raise ValueError("embedding dim must be a multiple of 4")
    lib = _L()
    with _on(h.device):
        scores = _empty(batch, dtype=torch.float32, device=h.device)
        rc = lib.distmult_fwd(_ptr(h), _ptr(h_idx), h.size(0), _ptr(t), _ptr(t_idx), t.size(0), _ptr(r), _ptr(r_idx),
                              r.size(0), batch, d, _ptr(scores), _stream())
    _lib.check(rc, "distmult_fwd")
    return scores


def check_indices(device=None) -> None:
    """Raise ``IndexError`` if any kernel since the last call met a head / tail / relation id outside its
    table (``rgcn_index_error_fetch``).  Such ids are never dereferenced - they are clamped to row 0 and a
    sticky device flag is raised - so, like torch's device-side assert on a bad index, the error surfaces
    at the next check (this call synchronises): the trainer and the evaluator check once per epoch / run."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    flag = ctypes.c_int(0)
    with _on(device):
        rc = _L().rgcn_index_error_fetch(ctypes.byref(flag), _stream())
    _lib.check(rc, "rgcn_index_error_fetch")
    if flag.value:
        raise IndexError("a head / tail / relation index was outside its embedding table "
                         "(clamped to row 0 on the device; results of that step are invalid)")


def segment_sum(rows: torch.Tensor, idx: torch.Tensor, num_rows: int) -> torch.Tensor:
    """``zeros(num_rows, d).index_add_(0, idx, rows)`` with a fixed summation order (deterministic; no
    atomics): autograd of ``table[idx]`` for a table of few rows (``rgcn_segment_sum``)."""
    _need_gpu("rows", rows, torch.float32)
    _need_gpu("idx", idx, torch.int64)
    if rows.dim() != 2 or idx.shape != (rows.size(0),) or rows.size(1) % 4:
        raise ValueError("rows must be [B, d] (d % 4 == 0) and idx [B]")
    lib = _L()
    b, d = rows.shape
    with _on(rows.device):
        out = _empty(num_rows, d, dtype=torch.float32, device=rows.device)
        nbytes = lib.rgcn_segment_sum_workspace_bytes(b, d, num_rows)
        ws = _workspace(nbytes, rows.device)
        rc = lib.rgcn_segment_sum(_ptr(rows), _ptr(idx), b, d, int(num_rows), _ptr(out), _ptr(ws), nbytes, _stream())
    _lib.check(rc, "rgcn_segment_sum")
    return out


def adam_clip_step(params, grads, exp_avgs, exp_avg_sqs, steps, lr: float, beta1: float, beta2: float, eps: float,
                   weight_decay: float = 0.0, adamw: bool = False, max_norm: float = 0.0,
                   total_norm: Optional[torch.Tensor] = None, amax_out=None) -> None:
    """``clip_grad_norm_(params, max_norm)`` (``max_norm <= 0``: no clipping) followed by one
    ``torch.optim.Adam`` / ``AdamW`` step, in two launches (``rgcn_adam_clip_step``).  All lists
    are parallel, fp32, contiguous CUDA tensors; ``steps[t]`` is the one-element device step count of
    tensor t (bumped here).  Parameters and moments are updated in place; gradients are left as
    they are (the clipped values are used, not stored).  ``amax_out``: per tensor an amax buffer (or None) that
    receives ``max |param|`` after the update (its written heads are cleared by the first launch; allocate it with
    ``amax_buffer``) - the scales the next step's transforms would otherwise scan for."""
    n = len(params)
    if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == len(steps) == n):
        raise ValueError("params, grads, exp_avgs, exp_avg_sqs and steps must be equally long")
    if n == 0:
        return
    if n > 32:
        raise ValueError("at most 32 parameter tensors per call")
    for i in range(n):
        for name, t in (("param", params[i]), ("grad", grads[i]), ("exp_avg", exp_avgs[i]),
                        ("exp_avg_sq", exp_avg_sqs[i]), ("step", steps[i])):
            _need_gpu(f"{name}[{i}]", t, torch.float32)
        if not (grads[i].shape == exp_avgs[i].shape == exp_avg_sqs[i].shape == params[i].shape) or steps[i].numel() != 1:
            raise ValueError(f"tensor {i}: param / grad / moments must have one shape, step one element")
    lib = _L()
    dev = params[0].device
    arr = ctypes.c_void_p * n
    numels = (ctypes.c_int64 * n)(*[p.numel() for p in params])
    amax_arr = None
    if amax_out is not None:
        if len(amax_out) != n:
            raise ValueError("amax_out must be as long as params")
        for a in amax_out:
            _check_amax("amax_out", a, dev)
        amax_arr = ctypes.cast(arr(*[_ptr(a) for a in amax_out]), ctypes.c_void_p)
    with _on(dev):
        nbytes = lib.rgcn_adam_workspace_bytes(n, ctypes.cast(numels, ctypes.c_void_p))
        ws = _workspace(nbytes, dev)
        rc = lib.rgcn_adam_clip_step(
            n, ctypes.cast(arr(*[_ptr(t) for t in params]), ctypes.c_void_p),
            ctypes.cast(arr(*[_ptr(t) for t in grads]), ctypes.c_void_p),
            ctypes.cast(arr(*[_ptr(t) for t in exp_avgs]), ctypes.c_void_p),
            ctypes.cast(arr(*[_ptr(t) for t in exp_avg_sqs]), ctypes.c_void_p),
            ctypes.cast(arr(*[_ptr(t) for t in steps]), ctypes.c_void_p), ctypes.cast(numels, ctypes.c_void_p),
            float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(adamw), float(max_norm),
            _ptr(total_norm), amax_arr, _ptr(ws), nbytes, _stream())
    _lib.check(rc, "rgcn_adam_clip_step")


def sample_batch(edge_index: torch.Tensor, edge_type: torch.Tensor, order: Optional[torch.Tensor],
                 cursor: Optional[torch.Tensor], batch: int, num_neg: int, num_nodes: int,
                 rng: Optional[torch.Tensor]):
    """One training mini-batch assembled on the device (``rgcn_sample_batch``): positives =
    columns ``order[cursor : cursor + batch]``, then ``num_neg`` corruptions of each (head or
    tail, fair coin, uniform replacement), then labels.  ``cursor`` (int64[1]) and ``rng``
    (int64[2] = seed, epoch) are DEVICE tensors.  -> (heads, tails, rels int64[B(1+k)], labels f32)."""
    _need_gpu("edge_index", edge_index, torch.int64)
    _need_gpu("edge_type", edge_type, torch.int64)
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_type.shape != (edge_index.size(1),):
        raise ValueError("edge_index must be [2, E] and edge_type [E]")
    e = edge_index.size(1)
    if order is not None:
        _need_gpu("order", order, torch.int64)
        if order.shape != (e,):
            raise ValueError(f"order must be [{e}]")
    if cursor is not None:
        _need_gpu("cursor", cursor, torch.int64)
        if cursor.numel() != 1:
            raise ValueError("cursor must hold one int64")
    if num_neg > 0:
        if rng is None:
            raise ValueError("rng (int64[2]: seed, epoch) is needed to draw negatives")
        _need_gpu("rng", rng, torch.int64)
        if rng.numel() != 2:
            raise ValueError("rng must hold two int64 (seed, epoch)")
    if batch < 0 or num_neg < 0 or (batch > 0 and e == 0):
        raise ValueError("batch / num_neg must be >= 0 and the graph must have columns")
    total = batch * (1 + num_neg)
    lib = _L()
    with _on(edge_index.device):
        heads = _empty(total, dtype=torch.int64, device=edge_index.device)
        tails, rels = torch.empty_like(heads), torch.empty_like(heads)
        labels = _empty(total, dtype=torch.float32, device=edge_index.device)
        rc = lib.rgcn_sample_batch(_ptr(edge_index), _ptr(edge_type), e, _ptr(order), _ptr(cursor), batch, num_neg,
                                   int(num_nodes), _ptr(rng), _ptr(heads), _ptr(tails), _ptr(rels), _ptr(labels),
                                   _stream())
    _lib.check(rc, "rgcn_sample_batch")
    return heads, tails, rels, labels


def distmult_bce_fwd(h, h_idx, t, t_idx, r, r_idx, labels, batch: int):
    """-> (scores [B], per-sample binary_cross_entropy_with_logits(scores, labels) [B]) in one launch."""
    d = h.size(1)
    for name, m, i in (("head", h, h_idx), ("tail", t, t_idx), ("rel", r, r_idx)):
        _check_operand(name, m, i, batch)
        if m.size(1) != d:
            raise ValueError("head / tail / relation embedding dims differ")
    if d % 4:
        raise ValueError("embedding dim must be a multiple of 4")
    _need_gpu("labels", labels, torch.float32)
    if labels.shape != (batch,):
        raise ValueError(f"labels must be [{batch}], got {tuple(labels.shape)}")
    lib = _L()
    with _on(h.device):
        scores = _empty(batch, dtype=torch.float32, device=h.device)
        loss = _empty(batch, dtype=torch.float32, device=h.device)
        rc = lib.distmult_bce_fwd(_ptr(h), _ptr(h_idx), h.size(0), _ptr(t), _ptr(t_idx), t.size(0), _ptr(r), _ptr(r_idx),
                                  r.size(0), _ptr(labels), batch, d, _ptr(scores), _ptr(loss), _stream())
    _lib.check(rc, "distmult_bce_fwd")
    return scores, loss


def basis_compose(comp: torch.Tensor, basis: torch.Tensor) -> torch.Tensor:
    """``(comp @ basis.view(B, -1)).view(R, d_in, d_out)`` - PyG's basis-decomposed relation weights (``rgcn_basis_compose``)."""
    _need_gpu("comp", comp, torch.float32)
    _need_gpu("basis", basis, torch.float32)
    if comp.dim() != 2 or basis.dim() != 3 or comp.size(1) != basis.size(0):
        raise ValueError("comp [R, B] and basis [B, d_in, d_out] expected")
    r, b = comp.shape
    inner = basis.size(1) * basis.size(2)
    lib = _L()
    with _on(comp.device):
        out = _empty((r, basis.size(1), basis.size(2)), dtype=torch.float32, device=comp.device)
        rc = lib.rgcn_basis_compose(_ptr(comp), _ptr(basis), r, b, inner, _ptr(out), _stream())
    _lib.check(rc, "rgcn_basis_compose")
    return out


def basis_compose_bwd(grad_weight: torch.Tensor, comp: torch.Tensor, basis: torch.Tensor, need_comp: bool = True,
                      need_basis: bool = True):
    """-> (grad_comp | None, grad_basis | None) of ``basis_compose``; deterministic (``rgcn_basis_compose_bwd``)."""
    _need_gpu("grad_weight", grad_weight, torch.float32)
    _need_gpu("comp", comp, torch.float32)
    _need_gpu("basis", basis, torch.float32)
    r, b = comp.shape
    inner = basis.size(1) * basis.size(2)
    if grad_weight.numel() != r * inner:
        raise ValueError("grad_weight must be [R, d_in, d_out]")
    lib = _L()
    with _on(comp.device):
        g_comp = _empty((r, b), dtype=torch.float32, device=comp.device) if need_comp else None
        g_basis = _empty(tuple(basis.shape), dtype=torch.float32, device=comp.device) if need_basis else None
        nbytes = lib.rgcn_basis_compose_bwd_workspace_bytes(r, b, inner) if need_comp else 0
        ws = _workspace(nbytes, comp.device) if need_comp else None
        rc = lib.rgcn_basis_compose_bwd(_ptr(grad_weight), _ptr(comp), _ptr(basis), r, b, inner, _ptr(g_comp), _ptr(g_basis),
                                        _ptr(ws), nbytes, _stream())
    _lib.check(rc, "rgcn_basis_compose_bwd")
    return g_comp, g_basis


def distmult_bce_reduce(loss: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor, loss_sum=None, correct=None,
                        cursor=None, cursor_add: int = 0) -> torch.Tensor:
    """-> ``mean(loss)`` as a one-element tensor, in ONE launch together with the epoch's device-resident running sums
    (``loss_sum`` float64 [] += mean * B, ``correct`` int64 [] += #{(scores > 0) == (labels > 0.5)}) and the batch cursor
    (``cursor`` int64 [1] += cursor_add) - ``src/train.py:300, 321-326``; any of the three may be None."""
    _need_gpu("loss", loss, torch.float32)
    b = loss.numel()
    if correct is not None:
        _need_gpu("scores", scores, torch.float32)
        _need_gpu("labels", labels, torch.float32)
        _need_gpu("correct", correct, torch.int64)
        if scores.numel() != b or labels.numel() != b:
            raise ValueError("loss, scores and labels must have one element per sample")
    if loss_sum is not None:
        _need_gpu("loss_sum", loss_sum, torch.float64)
    if cursor is not None:
        _need_gpu("cursor", cursor, torch.int64)
    lib = _L()
    with _on(loss.device):
        mean = _empty(1, dtype=torch.float32, device=loss.device)
        rc = lib.distmult_bce_reduce(_ptr(loss), _ptr(scores) if correct is not None else None,
                                     _ptr(labels) if correct is not None else None, b, _ptr(mean), _ptr(loss_sum),
                                     _ptr(correct), _ptr(cursor), int(cursor_add), _stream())
    _lib.check(rc, "distmult_bce_reduce")
    return mean


def _bwd_workspace(lib, batch: int, d: int, r, r_idx, device):
    nbytes = lib.distmult_bwd_workspace_bytes(batch, d, r.size(0) if r_idx is not None else 0)
    return _workspace(nbytes, device), nbytes


def distmult_bce_bwd(grad_mean_loss, scores, labels, h, h_idx, t, t_idx, r, r_idx, batch: int,
                     grad_h, grad_t, grad_r, zero_tables: bool = False) -> None:
    """Backward of ``mean(bce_with_logits(distmult(...), labels))``; ``grad_mean_loss`` is a
    one-element device tensor.  Writes like ``distmult_bwd`` (``zero_tables`` as there)."""
    _need_gpu("grad_mean_loss", grad_mean_loss, torch.float32)
    if grad_mean_loss.numel() != 1:
        raise ValueError("grad_mean_loss must hold one float")
    d = h.size(1)
    lib = _L()
    with _on(h.device):
        ws, nbytes = _bwd_workspace(lib, batch, d, r, r_idx, h.device)
        rc = lib.distmult_bce_bwd(_ptr(grad_mean_loss), _ptr(scores), _ptr(labels), _ptr(h), _ptr(h_idx), h.size(0),
                                  _ptr(t), _ptr(t_idx), t.size(0), _ptr(r), _ptr(r_idx), r.size(0), batch, d,
                                  _ptr(grad_h), _ptr(grad_t), _ptr(grad_r), _ptr(ws), nbytes, int(bool(zero_tables)),
                                  _stream())
    _lib.check(rc, "distmult_bce_bwd")


def distmult_bwd(gs, h, h_idx, t, t_idx, r, r_idx, batch: int, grad_h, grad_t, grad_r, zero_tables: bool = False) -> None:
    """Deterministic backward (no float atomics; two runs give the same bits): rows reached through an
    index vector are WRITTEN with the ordered sum of their samples' contributions; the rows nobody touches are
    cleared by the first launch itself (``zero_tables``: the buffers may then be ``torch.empty``) or must come
    zeroed from the caller; ``grad_h is grad_t`` (one table) is one key space."""
    _need_gpu("grad_scores", gs, torch.float32)
    d = h.size(1)
    lib = _L()
    with _on(h.device):
        ws, nbytes = _bwd_workspace(lib, batch, d, r, r_idx, h.device)
        rc = lib.distmult_bwd(_ptr(gs), _ptr(h), _ptr(h_idx), h.size(0), _ptr(t), _ptr(t_idx), t.size(0), _ptr(r),
                              _ptr(r_idx), r.size(0), batch, d, _ptr(grad_h), _ptr(grad_t), _ptr(grad_r), _ptr(ws),
                              nbytes, int(bool(zero_tables)), _stream())
    _lib.check(rc, "distmult_bwd")


def distmult_rank_tails(hr: torch.Tensor, emb: torch.Tensor, true_score: torch.Tensor,
                        tail: torch.Tensor) -> torch.Tensor:
    """``rank[b] = 1 + #{n != tail[b] : <hr[b], emb[n]> > true_score[b]}`` (int64 [B]): the
    rank of the true tail among all entities without the [B, N] score matrix
    (``evaluate.py:260-276``)."""
    _need_gpu("hr", hr, torch.float32)
    _need_gpu("emb", emb, torch.float32)
    _need_gpu("true_score", true_score, torch.float32)
    _need_gpu("tail", tail, torch.int64)
    b, d = hr.shape
    if emb.dim() != 2 or emb.size(1) != d or true_score.shape != (b,) or tail.shape != (b,):
        raise ValueError("hr [B, d], emb [N, d], true_score [B], tail [B] expected")
    if d % 32:
        raise ValueError("embedding dim must be a multiple of 32 for the fused ranking kernel")
    lib = _L()
    with _on(hr.device):
        beaten = torch.zeros(b, dtype=torch.int32, device=hr.device)
        rc = lib.distmult_rank_tails(_ptr(hr), _ptr(emb), _ptr(true_score), _ptr(tail), b, emb.size(0), d,
                                     _ptr(beaten), _stream())
    _lib.check(rc, "distmult_rank_tails")
    return beaten.to(torch.int64) + 1


def distmult_score_all_tails(head: torch.Tensor, rel: torch.Tensor, rel_idx: Optional[torch.Tensor],
                             emb: torch.Tensor):
    """``((head * rel[rel_idx]) @ emb.T, head * rel[rel_idx])`` - the [B, N] score matrix of
    ``LinkPredictor.score_all_tails`` (``rgcn.py:215-243``) from the ranking kernel's GEMM with a store epilogue, and
    the products it multiplied (the backward's operand).  ``rel_idx is None``: ``rel`` holds one row per b."""
    _need_gpu("head", head, torch.float32)
    _need_gpu("rel", rel, torch.float32)
    _need_gpu("emb", emb, torch.float32)
    b, d = head.shape
    if emb.dim() != 2 or emb.size(1) != d or rel.dim() != 2 or rel.size(1) != d:
        raise ValueError("head [B, d], rel [R, d], emb [N, d] expected")
    if rel_idx is not None:
        _need_gpu("rel_idx", rel_idx, torch.int64)
        if rel_idx.shape != (b,):
            raise ValueError("rel_idx [B] expected")
    elif rel.size(0) != b:
        raise ValueError("rel must hold one row per head row when no relation ids are given")
    if d % 32:
        raise ValueError("embedding dim must be a multiple of 32 for the score kernel")
    lib = _L()
    with _on(head.device):
        hr = _empty((b, d), dtype=torch.float32, device=head.device)
        scores = _empty((b, emb.size(0)), dtype=torch.float32, device=head.device)
        rc = lib.distmult_score_all_tails(_ptr(head), _ptr(rel), _ptr(rel_idx), rel.size(0), _ptr(emb), b, emb.size(0),
                                          d, _ptr(hr), _ptr(scores), _stream())
    _lib.check(rc, "distmult_score_all_tails")
    return scores, hr


# ----------------------------------------------------------------------------------
# Region: a pass (a fixed list of this library's launches) recorded once, then issued by ONE native call
# ----------------------------------------------------------------------------------
# The wrappers above cost 25-80 us of Python each (argument checks, ctypes marshalling, one torch.empty per output and
# workspace); an encoder step is ~14 of them: 0.54 ms of host time per eager step against 0.29 ms of kernels.  On a
# static graph the list of launches of a pass never changes - only the addresses of the step's tensors do - so a
# Region runs the pass's Python twice (once to learn the sizes of its allocations, once more with every allocation
# placed in ONE arena and every library call noted down with its arguments classified as constant / arena + offset /
# input k + offset), checks that issuing the noted list natively (``rgcn_sequence_run``) reproduces the recorded
# results bit for bit, and from then on a call is: one allocation, one C call, views for the tensors the caller looks
# at.  Any pass that does something a Region cannot note (a torch op in the middle, an entry point the native runner
# does not forward to, an allocation pattern that changes) keeps running through the wrappers - same kernels.
class _NotRecordable(Exception):
    pass


_SEQ_INDEX = {name: i for i, name in enumerate(_lib.SEQ_FUNCTIONS)}
_SEQ_STREAM_POS = {"rgcn_absmax": 5, "rgcn_absmax_multi": 6, "rgcn_absmax_pack": 13, "rgcn_weights_split_pack_multi": 12,
                   "rgcn_aggregate": 7, "rgcn_aggregate_and_reduce": 8, "rgcn_aggregate_amax": 9, "rgcn_aggregate_deferred": 8,
                   "rgcn_transform_fwd_split": 20, "rgcn_transform_bwd_input_split": 19, "rgcn_transform_first_split": 12,
                   "rgcn_transform_bwd_params_split_begin": 18, "rgcn_slab_reduce": 1, "rgcn_layer_fwd_fused": 17,
                   "rgcn_layer_bwd_input_fused": 17, "rgcn_transform_bwd_input_chain_split": 18}
_SEQ_PURE = ("rgcn_graph_tile_mask", "rgcn_graph_num_levels", "rgcn_graph_weight_bound", "rgcn_aggregate_deferrable",
             "rgcn_graph_num_edges", "rgcn_graph_num_nodes", "rgcn_graph_num_relations", "rgcn_abi_version", "rgcn_strerror")
REGIONS = True      # False: always through the wrappers (tests, tools/host_profile.py)


def guard_torch_op(what: str) -> None:
    """called where a pass falls back on a torch op: such a pass cannot be a Region"""
    if _REC is not None:
        raise _NotRecordable(what)


def _strides(shape):
    st, acc = [], 1
    for n in reversed(shape):
        st.append(acc)
        acc *= n
    return tuple(reversed(st))


class Lazy:
    """a tensor of a replayed pass that nobody has looked at yet: arena + offset (``.tensor()`` makes the view - one
    ``as_strided`` on the arena's float32 alias, which the pass's outputs share)"""
    __slots__ = ("arena", "offset", "shape", "dtype", "_t", "alias")

    def __init__(self, arena, offset, shape, dtype, alias=None):
        self.arena, self.offset, self.shape, self.dtype, self._t, self.alias = arena, offset, shape, dtype, None, alias

    def data_ptr(self) -> int:
        return self.arena.data_ptr() + self.offset

    def tensor(self) -> torch.Tensor:
        if self._t is None:
            if self.dtype == torch.float32 and self.alias is not None and self.offset % 4 == 0:
                self._t = torch.as_strided(self.alias, self.shape, _strides(self.shape), self.offset // 4)
            else:
                n = 1
                for s in self.shape:
                    n *= s
                nbytes = n * torch.empty((), dtype=self.dtype).element_size() if n else 0
                self._t = self.arena[self.offset: self.offset + nbytes].view(self.dtype).view(self.shape)
        return self._t


def materialize(t):
    return t.tensor() if isinstance(t, Lazy) else t


def _signature(t):
    """(dtype, shape, device index) of a pass's input - a tensor or the arena view of an earlier pass - or None"""
    if t is None:
        return None
    if isinstance(t, Lazy):
        return (t.dtype, tuple(t.shape), t.arena.get_device())
    return (t.dtype, tuple(t.shape), t.get_device())           # (-1 on the host)


class _Proxy:
    def __init__(self, rec):
        self._rec, self._real = rec, _lib.load()

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _SEQ_PURE or name.endswith(("_bytes", "_supported")):
            return fn
        if name not in _SEQ_INDEX:
            raise _NotRecordable(f"{name} is not an entry point the native runner forwards to")
        rec = self._rec

        def call(*args):
            rc = fn(*args)
            rec.note(name, args)
            return rc
        return call


class _Recorder:
    ALIGN = 256

    def __init__(self, device, sizes=None, inputs=()):
        self.device, self.sizes, self.proxy = device, ([] if sizes is None else sizes), None
        self.recording = sizes is not None
        self.calls, self.jobs, self.cursor = [], {}, 0
        self.keep = []                                      # sizing run: allocations stay alive (no address reuse)
        if self.recording:
            self.offsets, total = [], 0
            for nb in sizes:
                self.offsets.append(total)
                total += (nb + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            # zero filled, like the arena of the check replay: bytes no launch writes (the unused entries of an amax
            # buffer, alignment gaps of the split weight images) then compare equal
            self.arena = torch.zeros((max(total, self.ALIGN) + 3) // 4 * 4, dtype=torch.uint8, device=device)
            self.arena_bytes = (max(total, self.ALIGN) + 3) // 4 * 4
            self.base = self.arena.data_ptr()
            self.inputs = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) if t is not None else None
                           for t in inputs]
            self.input_signature = [_signature(t) for t in inputs]
            self.proxy = _Proxy(self)

    def alloc(self, shape, dtype, device):
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if not self.recording:
            self.sizes.append(nbytes)
            t = torch.empty(shape, dtype=dtype, device=device)
            self.keep.append(t)
            return t
        k = self.cursor
        if k >= len(self.sizes) or self.sizes[k] != nbytes or torch.device(device) != self.arena.device:
            raise _NotRecordable("the pass allocates differently from its first run")
        self.cursor += 1
        if nbytes == 0:
            return torch.empty(shape, dtype=dtype, device=device)
        off = self.offsets[k]
        return self.arena[off: off + nbytes].view(dtype).view(shape)

    def _classify(self, ptr: int):
        if not ptr:
            return (_lib.SEQ_IMM, 0, 0)
        if self.base <= ptr < self.base + self.arena_bytes:
            return (_lib.SEQ_BASE, 0, ptr - self.base)
        for k, span in enumerate(self.inputs):
            if span is not None and span[0] <= ptr < span[1]:
                return (_lib.SEQ_BASE, k + 1, ptr - span[0])
        return (_lib.SEQ_IMM, 0, ptr)                       # static: graph structures, handles, masks

    @staticmethod
    def _as_int(a) -> int:
        if a is None:
            return 0
        if isinstance(a, ctypes.c_void_p):
            return a.value or 0
        return int(a)

    def note(self, name, args):
        proto = _lib.PROTOTYPES[name][1]
        if len(args) != len(proto):
            raise _NotRecordable(f"{name}: argument count")
        arrays = _lib.SEQ_HOST_ARRAYS.get(name, {})
        descs = []
        for i, (a, ty) in enumerate(zip(args, proto)):
            if i == _SEQ_STREAM_POS[name]:
                descs.append((_lib.SEQ_STREAM, 0, 0))
            elif i in arrays:
                is_ptr, cnt_pos = arrays[i]
                addr, n = self._as_int(a), int(args[cnt_pos])
                if not addr:
                    descs.append((_lib.SEQ_IMM, 0, 0))
                    continue
                raw = (ctypes.c_uint64 * n).from_address(addr)
                descs.append(("array", [self._classify(int(v)) if is_ptr else (_lib.SEQ_IMM, 0, int(v)) for v in raw]))
            elif ty is ctypes.c_float:
                descs.append((_lib.SEQ_FLOAT, 0, float(a)))
            elif ty is ctypes.POINTER(_lib.SlabJob):
                if a is None:
                    descs.append((_lib.SEQ_IMM, 0, 0))
                else:
                    job = a._obj
                    slot = self.jobs.setdefault(id(job), len(self.jobs))
                    self.keep.append(job)
                    descs.append((_lib.SEQ_JOB, slot, 0))
            elif ty is ctypes.c_void_p:
                descs.append(self._classify(self._as_int(a)))
            else:
                descs.append((_lib.SEQ_IMM, 0, int(a)))
        self.calls.append((_SEQ_INDEX[name], descs))


class _Plan:
    def __init__(self, rec: _Recorder, outputs, want=()):
        # Outputs the caller LOOKS AT (`want`) and that are whole allocations of the pass get a tensor of their own at
        # replay time, addressed through one more base pointer: an arena view handed out of an autograd Function would
        # refuse in-place ops on it ("a view ... is being modified inplace"), would keep the whole arena (aggregates,
        # workspaces, split images) alive for as long as the caller holds it, and `torch.save` would write all of it.
        self.external = []                                   # (arena offset, nbytes, shape, dtype) per such output
        ext_of = {}
        n_in = len(rec.inputs)
        spans = {off: nb for off, nb in zip(rec.offsets, rec.sizes)}
        outputs = list(outputs)
        for i in sorted(want):
            spec = outputs[i] if i < len(outputs) else None
            if spec is None or spec[0] != "arena":
                continue
            off, shape, dtype = spec[1], spec[2], spec[3]
            n = 1
            for d in shape:
                n *= d
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            if nbytes and spans.get(off) == nbytes:
                if off not in ext_of:
                    ext_of[off] = len(self.external)
                    self.external.append((off, nbytes, shape, dtype))
                outputs[i] = ("external", ext_of[off])

        def rebase(kind, index, value):
            if kind == _lib.SEQ_BASE and index == 0:
                for e, (off, nbytes, _, _) in enumerate(self.external):
                    if off <= value < off + nbytes:
                        return (kind, 1 + n_in + e, value - off)
            return (kind, index, value)

        flat, calls = [], []
        tails = []                                           # array entries live behind the calls' own arguments
        for fn, descs in rec.calls:
            first = len(flat)
            for d in descs:
                if d[0] == "array":
                    flat.append(["array", d[1]])
                else:
                    flat.append(list(d))
            calls.append((fn, len(descs), first))
        for slot in flat:
            if slot[0] == "array":
                entries = slot[1]
                start = len(flat) + len(tails)
                tails.extend(entries)
                slot[:] = [_lib.SEQ_ARRAY, start, len(entries)]
        allargs = [rebase(*a) for a in flat] + [rebase(*t) for t in tails]
        self.num_calls, self.num_args = len(calls), len(allargs)
        self.calls = (_lib.SeqCall * max(1, len(calls)))(*[_lib.SeqCall(fn, n, first) for fn, n, first in calls])
        args = (_lib.SeqArg * max(1, len(allargs)))()
        for i, (kind, index, value) in enumerate(allargs):
            args[i].kind, args[i].index = kind, index
            if kind == _lib.SEQ_FLOAT:
                args[i].value = ctypes.c_int64.from_buffer_copy(ctypes.c_double(value)).value
            else:
                args[i].value = value
        self.args = args
        self.arena_bytes, self.device = rec.arena_bytes, rec.device
        self.outputs = outputs                               # per output: ("arena", off, shape, dtype) | ("input", k) | None
        self.num_jobs = len(rec.jobs)
        # external outputs under 256 KB each (bias / root / weight gradients) are views of one slab of their own - one
        # allocation instead of six; the large ones (layer outputs, the input gradient) stay single tensors
        self._small_off, self._small_bytes = [], 0
        for _, nbytes, _, _ in self.external:
            if nbytes <= (256 << 10):
                self._small_off.append(self._small_bytes)
                self._small_bytes += (nbytes + 255) // 256 * 256
            else:
                self._small_off.append(None)
        self._small = sum(o is not None for o in self._small_off) >= 2
        self._small_f32 = all(dt == torch.float32 for o, (_, _, _, dt) in zip(self._small_off, self.external) if o is not None)

        def dense_strides(shape):
            st, acc = [], 1
            for dim in reversed(shape):
                st.append(acc)
                acc *= dim
            return tuple(reversed(st))
        self._small_strides = [dense_strides(shape) for _, _, shape, _ in self.external]
        # what the recorded addresses stand for: a later call whose inputs differ in type, shape or place must not be
        # replayed (the wrappers would have refused it; the native list would read the wrong bytes)
        self.signature = rec.input_signature

    def matches(self, inputs) -> bool:
        """the inputs of this call are what the recorded addresses stand for (a few attribute reads per input: this runs
        on every replay)"""
        sigs = self.signature
        if len(inputs) != len(sigs):
            return False
        for t, sig in zip(inputs, sigs):
            if t is None or sig is None:
                if t is not sig:                             # (both None, or a mismatch)
                    return False
                continue
            if type(t) is Lazy:                              # (an arena view is dense by construction)
                if t.dtype is not sig[0] or t.shape != sig[1] or t.arena.get_device() != sig[2]:
                    return False
            elif (t.dtype is not sig[0] or t.shape != sig[1] or t.get_device() != sig[2]
                  or not t.is_contiguous()):                 # (a strided input would be read as dense through its data_ptr)
                return False
        return True

    def run(self, inputs, want, fill=None):
        """-> list of outputs: tensors for the positions in `want`, Lazy for the rest.  `fill`: a byte value the arena
        and the external outputs are filled with first (the acceptance replays of Region.run); None = torch.empty"""
        lib = _lib.load()
        with _on(self.device):
            if fill is None:
                arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device)
                if self._small:                              # the small ones (parameter gradients) share ONE allocation
                    if self._small_f32:                      # all float32: one as_strided per output instead of slice + 2 views
                        slab = torch.empty(self._small_bytes // 4, dtype=torch.float32, device=self.device)
                        ext = [slab.as_strided(shape, st, o // 4) if o is not None else
                               torch.empty(shape, dtype=dtype, device=self.device)
                               for o, st, (_, nb, shape, dtype) in zip(self._small_off, self._small_strides, self.external)]
                    else:
                        slab = torch.empty(self._small_bytes, dtype=torch.uint8, device=self.device)
                        ext = [slab[o: o + nb].view(dtype).view(shape) if o is not None else
                               torch.empty(shape, dtype=dtype, device=self.device)
                               for o, (_, nb, shape, dtype) in zip(self._small_off, self.external)]
                else:
                    ext = [torch.empty(shape, dtype=dtype, device=self.device) for _, _, shape, dtype in self.external]
            else:
                arena = torch.full((self.arena_bytes,), fill, dtype=torch.uint8, device=self.device)
                ext = [torch.full((nbytes,), fill, dtype=torch.uint8, device=self.device).view(dtype).view(shape)
                       for _, nbytes, shape, dtype in self.external]
            nb = len(inputs) + 1 + len(ext)
            bases = (ctypes.c_void_p * nb)(arena.data_ptr(), *[t.data_ptr() if t is not None else None for t in inputs],
                                           *[t.data_ptr() for t in ext])
            rc = lib.rgcn_sequence_run(self.calls, self.num_calls, self.args, self.num_args, bases, nb, _stream())
        _lib.check(rc, "rgcn_sequence_run")
        outs = []
        alias = arena.view(torch.float32) if want else None          # one alias for all the float32 views of this pass
        for i, spec in enumerate(self.outputs):
            if spec is None:
                outs.append(None)
            elif spec[0] == "input":
                outs.append(inputs[spec[1]])
            elif spec[0] == "external":
                outs.append(ext[spec[1]])
            else:
                lz = Lazy(arena, spec[1], spec[2], spec[3], alias)
                outs.append(lz.tensor() if i in want else lz)
        return outs


class Region:
    """``Region(name, fn)``: ``fn(*tensors, **static) -> tuple of tensors | None``, a pass made of library calls and
    ``_empty`` allocations only.  ``run(owner, key, tensors, static, want)``: through the wrappers three times, natively from
    then on; plans live on ``owner`` (the bucketed graph: they hold addresses of its structures) and die with it."""

    DISABLED = "disabled"

    def __init__(self, name: str, fn):
        self.name, self.fn = name, fn

    def run(self, owner, key, tensors, static, want=()):
        global _REC
        tensors = list(tensors)
        if not REGIONS or GATHER_EVENTS is not None or GEMM_EVENTS is not None or FUSED_EVENTS is not None or _REC is not None:
            return self._eager(tensors, static)
        store = owner.__dict__.setdefault("_regions", {})
        full_key = (self.name, key, GEMM_PRECISION)
        state = store.get(full_key)
        if isinstance(state, _Plan):
            if state.matches(tensors):
                try:
                    return state.run(tensors, want)
                except torch.cuda.OutOfMemoryError:          # the arena lays a pass's temporaries side by side: a step that
                    store[full_key] = self.DISABLED          # fits through the wrappers (which free as they go) keeps running
                    return self._eager(tensors, static)
            return self._eager(tensors, static)              # (the wrappers say what is wrong with these inputs)
        if state == self.DISABLED or torch.cuda.is_current_stream_capturing():
            return self._eager(tensors, static)
        if any(isinstance(t, Lazy) for t in tensors):
            tensors = [materialize(t) for t in tensors]
        if any(t is not None and not t.is_contiguous() for t in tensors):
            return self._eager(tensors, static)
        dev = next(t.device for t in tensors if t is not None)
        if state is None:                                    # first run: plain (lazily built structures come into being)
            store[full_key] = "warm"
            return self._eager(tensors, static)
        if state == "warm":                                  # second run: learn the allocation sizes
            rec = _Recorder(dev)
            _REC = rec
            try:
                outs = self.fn(*tensors, **static)
            except _NotRecordable:
                _REC = None
                store[full_key] = self.DISABLED
                return self._eager(tensors, static)
            finally:
                _REC = None
            store[full_key] = ("sizes", rec.sizes)
            return list(outs)
        # third run: every allocation in one arena, every call noted; then the native replay must reproduce it
        try:
            rec = _Recorder(dev, sizes=state[1], inputs=tensors)
        except torch.cuda.OutOfMemoryError:
            store[full_key] = self.DISABLED
            return self._eager(tensors, static)
        _REC = rec
        try:
            outs = list(self.fn(*tensors, **static))
        except _NotRecordable:
            store[full_key] = self.DISABLED
            _REC = None
            return self._eager(tensors, static)
        finally:
            _REC = None
        specs = []
        for o in outs:
            if o is None:
                specs.append(None)
                continue
            kind = rec._classify(o.data_ptr()) if o.numel() else (_lib.SEQ_IMM, 0, 0)
            if kind[0] == _lib.SEQ_BASE and kind[1] == 0 and o.is_contiguous():
                specs.append(("arena", kind[2], tuple(o.shape), o.dtype))
            elif kind[0] == _lib.SEQ_BASE and kind[2] == 0 and tensors[kind[1] - 1] is o:
                specs.append(("input", kind[1] - 1))
            else:
                store[full_key] = self.DISABLED              # an output the record cannot place
                return outs
        if rec.cursor != len(rec.sizes) or len(rec.jobs) > 8:
            store[full_key] = self.DISABLED
            return outs
        plan = _Plan(rec, specs, want)
        # Acceptance: the noted list, issued natively, must reproduce the recorded results bit for bit - replayed TWICE,
        # into memory filled with 0x00 (what the recording ran in: bytes no launch writes compare equal) and with 0xFF
        # (what a production replay may meet: torch.empty).  A byte of an output that differs between the two replays
        # must be one NO launch defines (0x00 in the one, 0xFF in the other: the unused entries of an amax buffer, the
        # alignment gaps of the split images); anything else means a launch READS memory it expects cleared, which the
        # zero-filled check alone would have passed and torch.empty would break.
        def as_bytes(t):
            return t.contiguous().view(-1).view(torch.uint8)
        try:
            everything = set(range(len(specs)))
            zeros = plan.run(tensors, want=everything, fill=0)
            ones = plan.run(tensors, want=everything, fill=255)
            same = True
            for a, z, f in zip(outs, zeros, ones):
                if a is None or a is z:
                    same = same and (z is None or a is z)
                    continue
                za, fa = as_bytes(z), as_bytes(f)
                undefined = za != fa
                same = same and torch.equal(a, z) and bool(((za == 0) & (fa == 255))[undefined].all())
        except (RuntimeError, ValueError, IndexError):
            same = False
        store[full_key] = plan if same else self.DISABLED
        # this run's results sit in the recorder's arena: what the caller looks at leaves it as a tensor of its own, like
        # the outputs of every later (replayed) and earlier (eager) step - no view of a shared buffer is handed out
        return [o.clone() if (i in want and specs[i] is not None and specs[i][0] == "arena") else o for i, o in enumerate(outs)]

    def _eager(self, tensors, static):
        return list(self.fn(*[materialize(t) for t in tensors], **static))

"""CPU oracle for the R-GCN layer + DistMult head.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package
(``primekg_rgcn_linkprediction_amd``) never does: its compute path is the HIP
library and it raises when that library is missing.

What is restated here
---------------------
The reference's layer arithmetic is *not* in the reference tree: it imports
``torch_geometric.nn.RGCNConv`` (``src/models/rgcn.py:17``; constructed at
``rgcn.py:72-85``, called at ``rgcn.py:123`` and ``rgcn.py:128``).  The
dependency is ``torch-geometric>=2.4.0`` (``requirements.txt:2``, no exact pin,
no lock file) and is absent from this image and from the GPU box.  This file
restates the published algorithm of PyG's pure-PyTorch path (no ``pyg_lib``):

    out = zeros(N, d_out)
    W   = weight                              (or comp @ weight.view(B, -1))
    for r in range(R):
        cols = edge_index[:, edge_type == r]  # order preserving
        x_j  = x.index_select(0, cols[0])     # source = row 0
        cnt  = zeros(N).scatter_add_(0, cols[1], 1).clamp(min=1)
        s    = zeros(N, d_in).scatter_add_(0, cols[1], x_j)   # target = row 1
        out  = out + (s / cnt[:, None]) @ W[r]
    out = out + x @ root
    out = out + bias

Parity status: **the layer is "parity unpinned"** - the reference holds no
numeric golden vector for it (its own tests assert shapes only,
``rgcn.py:450-451, 489-492, 550-557``) and PyG cannot be run here.  It is
anchored instead by (a) the independent float64 dense formulation below,
(b) the parameter count 2,078,208 (``results_final/results.json:28``) which
fixes every parameter shape, and (c) the reference's own DistMult head and
model wiring run in this container (``tests/golden/make_golden.py``), which
*are* pinned by reference-run outputs.

The DistMult head follows ``rgcn.py:189-213`` (forward) and ``rgcn.py:215-243``
(score_all_tails); the encoder wiring follows ``rgcn.py:97-130``.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# parameter init (PyG ``nn.inits.glorot`` / ``zeros``)
# --------------------------------------------------------------------------
def glorot_(t: Optional[torch.Tensor]) -> None:
    """U(-a, a), a = sqrt(6 / (size(-2) + size(-1))); no-op for ``None``."""
    if t is not None:
        a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
        t.data.uniform_(-a, a)


# --------------------------------------------------------------------------
# restatement #1: op-for-op loop path (what PyG executes without pyg_lib)
# --------------------------------------------------------------------------
def mean_aggregate_ref(x: torch.Tensor, edge_index: torch.Tensor,
                       edge_type: torch.Tensor, num_relations: int) -> torch.Tensor:
    """agg[N, R, d_in]: per-(destination, relation) mean of source rows.

    Row A4 of SURVEY section 8a.  Empty (i, r) gives exactly 0 (count clamped
    to 1); duplicate edge columns count once each; no self loops are added.
    """
    n, d = x.shape
    agg = x.new_zeros(n, num_relations, d)
    for r in range(num_relations):
        cols = edge_index[:, edge_type == r]
        x_j = x.index_select(0, cols[0])
        cnt = x.new_zeros(n).scatter_add_(0, cols[1], x.new_ones(cols.size(1)))
        cnt = cnt.clamp(min=1)
        s = x.new_zeros(n, d).scatter_add_(0, cols[1].view(-1, 1).expand(-1, d), x_j)
        agg[:, r, :] = s / cnt.view(-1, 1)
    return agg


def effective_weight(weight: torch.Tensor, comp: Optional[torch.Tensor],
                     num_relations: int) -> torch.Tensor:
    """``weight`` itself, or the basis composition ``comp @ weight.view(B,-1)``."""
    if comp is None:
        return weight
    b, d_in, d_out = weight.shape
    return (comp @ weight.view(b, -1)).view(num_relations, d_in, d_out)


def rgcn_conv_ref(x, edge_index, edge_type, weight, root, bias, comp=None,
                  num_relations: Optional[int] = None) -> torch.Tensor:
    """Forward of the layer in the exact op order of PyG's loop path."""
    if num_relations is None:
        num_relations = comp.size(0) if comp is not None else weight.size(0)
    n = x.size(0)
    w = effective_weight(weight, comp, num_relations)
    out = torch.zeros(n, w.size(-1), device=x.device)   # fp32 zeros (checklist item 6)
    for r in range(num_relations):
        cols = edge_index[:, edge_type == r]
        x_j = x.index_select(0, cols[0])
        cnt = x.new_zeros(n).scatter_add_(0, cols[1], x.new_ones(cols.size(1)))
        cnt = cnt.clamp(min=1)
        s = x.new_zeros(n, x.size(1)).scatter_add_(
            0, cols[1].view(-1, 1).expand(-1, x.size(1)), x_j)
        h = s / cnt.view(-1, 1)
        out = out + (h @ w[r])
    if root is not None:
        out = out + x @ root
    if bias is not None:
        out = out + bias
    return out


class RGCNConvRef(nn.Module):
    """nn.Module wrapper with PyG's constructor signature, parameter names,
    shapes and init order (weight, comp, root, bias)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases=None,
                 num_blocks=None, aggr="mean", root_weight=True, is_sorted=False,
                 bias=True):
        super().__init__()
        if num_bases is not None and num_blocks is not None:
            raise ValueError("Can not apply both basis-decomposition and "
                             "block-diagonal-decomposition at the same time.")
        if num_blocks is not None or aggr != "mean":
            raise NotImplementedError
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases = num_relations, num_bases
        if num_bases is not None:
            self.weight = nn.Parameter(torch.empty(num_bases, in_channels[0], out_channels))
            self.comp = nn.Parameter(torch.empty(num_relations, num_bases))
        else:
            self.weight = nn.Parameter(torch.empty(num_relations, in_channels[0], out_channels))
            self.register_parameter("comp", None)
        if root_weight:
            self.root = nn.Parameter(torch.empty(in_channels[1], out_channels))
        else:
            self.register_parameter("root", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight)
        glorot_(self.comp)
        glorot_(self.root)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, edge_index, edge_type=None):
        assert edge_type is not None
        return rgcn_conv_ref(x, edge_index, edge_type, self.weight, self.root,
                             self.bias, self.comp, self.num_relations)


# --------------------------------------------------------------------------
# restatement #2: independent float64 dense formulation (small N only)
# --------------------------------------------------------------------------
def rgcn_conv_dense_f64(x, edge_index, edge_type, weight, root, bias, comp=None,
                        num_relations: Optional[int] = None) -> torch.Tensor:
    """out = sum_r D_r^-1 A_r x W_r + x root + b with dense count matrices
    A_r[i, j] = #edges j->i of type r.  O(R N^2) memory: N <= ~2k."""
    if num_relations is None:
        num_relations = comp.size(0) if comp is not None else weight.size(0)
    n = x.size(0)
    xd = x.double()
    w = effective_weight(weight.double(), None if comp is None else comp.double(),
                         num_relations)
    out = torch.zeros(n, w.size(-1), dtype=torch.float64)
    src, dst = edge_index[0].numpy(), edge_index[1].numpy()
    et = edge_type.numpy()
    for r in range(num_relations):
        a = np.zeros((n, n), dtype=np.float64)
        m = et == r
        np.add.at(a, (dst[m], src[m]), 1.0)
        deg = np.maximum(a.sum(axis=1), 1.0)
        a = torch.from_numpy(a / deg[:, None])
        out += (a @ xd) @ w[r]
    if root is not None:
        out += xd @ root.double()
    if bias is not None:
        out += bias.double()
    return out


# --------------------------------------------------------------------------
# relation bucketing (integer work: bit exact)
# --------------------------------------------------------------------------
def bucket_ref(edge_index: torch.Tensor, edge_type: torch.Tensor, num_nodes: int,
               num_relations: int, transpose: bool = False):
    """CSR-by-relation restatement of ``edge_index[:, edge_type == r]`` for every
    r at once (row A2).  Segment id = node * R + rel, where node is the
    destination (forward structure) or the source (``transpose=True``, the
    structure backward needs, row A7).  A *stable* sort keeps the reference's
    order-preserving column selection inside every segment.

    Returns numpy arrays: rowptr int32[N*R+1], col int32[E] (the other
    endpoint), perm int64[E] (original column of each bucketed edge) and
    cnt float32[N*R] (max(1, segment size)).
    """
    ei = edge_index.numpy().astype(np.int64)
    et = edge_type.numpy().astype(np.int64)
    e = ei.shape[1]
    if e and (ei.min() < 0 or ei.max() >= num_nodes or et.min() < 0 or
              et.max() >= num_relations):
        raise ValueError("edge_index / edge_type out of range")
    key_node, other = (ei[0], ei[1]) if transpose else (ei[1], ei[0])
    key = key_node * num_relations + et
    perm = np.argsort(key, kind="stable").astype(np.int64)
    deg = np.bincount(key, minlength=num_nodes * num_relations)
    rowptr = np.zeros(num_nodes * num_relations + 1, dtype=np.int32)
    np.cumsum(deg, out=rowptr[1:])
    col = other[perm].astype(np.int32)
    cnt = np.maximum(deg, 1).astype(np.float32)
    return rowptr, col, perm, cnt


# --------------------------------------------------------------------------
# DistMult head (rgcn.py:189-243) and encoder wiring (rgcn.py:97-130)
# --------------------------------------------------------------------------
def distmult_ref(head_emb, tail_emb, rel_emb_rows):
    """scores[b] = sum_d h * r * t   (rgcn.py:211); r already gathered/dropped."""
    return torch.sum(head_emb * rel_emb_rows * tail_emb, dim=1)


def distmult_all_tails_ref(head_emb, rel_emb_rows, all_emb):
    """(h * r) @ E^T   (rgcn.py:238-241)."""
    return (head_emb * rel_emb_rows) @ all_emb.t()


def cosine_scores_ref(embeddings, drug_indices, disease_indices):
    """numpy float32, as ``src/compare_methods.py:384-397`` (RGCNMethod.predict_all):
    normalise rows, dot, map [-1, 1] -> [0, 1].  -> [n_drug, n_disease]"""
    import numpy as np
    emb = np.asarray(embeddings, dtype=np.float32)
    a, b = emb[np.asarray(drug_indices)], emb[np.asarray(disease_indices)]
    a = a / np.linalg.norm(a, axis=1, keepdims=True)
    b = b / np.linalg.norm(b, axis=1, keepdims=True)
    return (np.dot(a, b.T) + 1) / 2


def top_drugs_ref(embeddings, disease_idx, drug_indices, top_k=10, threshold=0.0):
    """as ``src/case_studies.py:236-284``: scores of all candidate drugs, filter by threshold,
    stable sort by score descending, first top_k.  -> [(drug_idx, score)]"""
    scores = cosine_scores_ref(embeddings, drug_indices, [disease_idx])[:, 0]
    predictions = [(int(drug_indices[i]), float(scores[i])) for i in range(len(scores)) if scores[i] >= threshold]
    predictions.sort(key=lambda p: p[1], reverse=True)
    return predictions[:top_k]


def encoder_ref(emb_weight, conv1: dict, conv2: dict, edge_index, edge_type,
                dropout_p: float = 0.0, training: bool = False, relu_mask=None):
    """conv1 -> relu -> dropout -> conv2 (rgcn.py:117-130); convN are dicts of
    weight/root/bias[/comp] tensors.  ``relu_mask`` (bool [N, hidden], optional): the ReLU decisions
    to use instead of this evaluation's own ``z > 0`` - a full-size comparison passes the device's, so
    that a pre-activation within fp32 rounding of zero cannot switch a unit's whole gradient on one
    side only (the forward changes by at most that rounding)."""
    x = rgcn_conv_ref(emb_weight, edge_index, edge_type, conv1["weight"], conv1["root"],
                      conv1["bias"], conv1.get("comp"))
    x = F.relu(x) if relu_mask is None else x * relu_mask.to(x.dtype)
    x = F.dropout(x, dropout_p, training)
    return rgcn_conv_ref(x, edge_index, edge_type, conv2["weight"], conv2["root"],
                         conv2["bias"], conv2.get("comp"))


# --------------------------------------------------------------------------
# restatement #3: the two-layer encoder with its backward spelled out, in float64
# --------------------------------------------------------------------------
def _r16(t: torch.Tensor) -> torch.Tensor:
    """round to fp16 (nearest even) and come back in the input dtype"""
    return t.to(torch.float32).half().to(t.dtype)


def _r16_scaled(t: torch.Tensor) -> torch.Tensor:
    """round to fp16 under the tensor's power-of-two scale (its largest magnitude scaled into [2^14, 2^15)):
    what the one-pass gradient GEMMs of configs[4] do to an operand - loss scaling per tensor, so that
    gradients of ~1e-7 neither underflow nor lose bits"""
    amax = float(t.abs().max())
    if amax == 0.0 or not math.isfinite(amax):
        return t
    e = 14 - math.floor(math.log2(amax))
    return _r16(t * 2.0 ** e) * 2.0 ** (-e)


def _segment_counts(edge_index, edge_type, n, r, dtype):
    key = edge_index[1] * r + edge_type
    return torch.bincount(key, minlength=n * r).clamp(min=1).to(dtype)          # cnt[dst * R + rel]


def _mean_agg(x, edge_index, edge_type, n, r, cnt):
    """[N, R*d]: per-(dst, rel) mean of source rows (rows A3 + A4), any float dtype"""
    d = x.size(1)
    key = edge_index[1] * r + edge_type
    s = x.new_zeros(n * r, d).index_add_(0, key, x.index_select(0, edge_index[0]))
    return (s / cnt.view(-1, 1)).view(n, r * d)


def _mean_agg_transposed(g, edge_index, edge_type, n, r, cnt):
    """autograd of ``_mean_agg``: out[src, rel] += g[dst] / cnt[dst, rel]  -> [N, R*d]"""
    d = g.size(1)
    w = 1.0 / cnt[edge_index[1] * r + edge_type]
    rows = g.index_select(0, edge_index[1]) * w.view(-1, 1)
    return g.new_zeros(n * r, d).index_add_(0, edge_index[0] * r + edge_type, rows).view(n, r * d)


def encoder_explicit_f64(emb, conv1: dict, conv2: dict, edge_index, edge_type, cot,
                         relu_mask: Optional[torch.Tensor] = None, half_forward: bool = False,
                         half_backward: bool = False):
    """conv1 -> relu -> conv2 (rgcn.py:117-130) and its whole backward evaluated in float64,
    every step written out (no autograd), from float32 parameters.

    ``half_forward=False``: the float64 evaluation of the formula the loop path above computes -
    the yardstick the fp32 results (the HIP path's and restatement #1's) are measured against.

    ``half_forward=True``: the EXACT MEANING of BASELINE configs[4] ("fp16 features + fp32
    accumulate") as the HIP path defines it: each gather reads the fp16-rounded feature table,
    each transform rounds both operands ``[agg | x]`` and ``[W ; root]`` to fp16 (nearest even),
    products are exact and sums wide; the backward is the fp32 formula on the tensors the forward
    saved (un-rounded ``agg``, ``x``, ``h``, fp32 weights).  ``half_backward=True`` additionally
    rounds the operands of the three gradient GEMMs per layer to fp16, each under its tensor's
    power-of-two scale (``_r16_scaled``: gradient tables, their aggregates, ``[agg | x]`` and the weights),
    sums wide; the gradient gathers stay exact.

    ``relu_mask`` [N, hidden] (bool): the ReLU decisions to use in the backward (pass the
    device's ``h > 0`` so that a pre-activation within rounding of zero cannot flip a unit
    between the two sides); default ``h > 0`` of this evaluation.

    -> dict(out, h, grads={"emb", "conv1.weight", ..., "conv2.bias"[, "convN.comp"]})"""
    f64 = torch.float64
    n, r = emb.size(0), (conv1["comp"].size(0) if conv1.get("comp") is not None else conv1["weight"].size(0))
    cnt = _segment_counts(edge_index, edge_type, n, r, f64)
    rf = _r16 if half_forward else (lambda t: t)
    rb = _r16_scaled if half_backward else (lambda t: t)

    def weights(c):
        w = effective_weight(c["weight"].to(f64), None if c.get("comp") is None else c["comp"].to(f64), r)
        wcat = w.reshape(-1, w.size(-1))
        if c.get("root") is not None:
            wcat = torch.cat([wcat, c["root"].to(f64)])
        return w, wcat

    def layer_fwd(x, c):
        _, wcat = weights(c)
        agg = _mean_agg(rf(x), edge_index, edge_type, n, r, cnt)
        a = torch.cat([agg, x], 1) if c.get("root") is not None else agg
        z = rf(a) @ rf(wcat)
        if c.get("bias") is not None:
            z = z + c["bias"].to(f64)
        return agg, a, z

    def layer_bwd(g, a, c, want_x=True):
        w, wcat = weights(c)
        d_in = w.size(1)
        gwcat = rb(a).t() @ rb(g)
        gw_eff = gwcat[: r * d_in].view(r, d_in, -1)
        out = {"root": gwcat[r * d_in:] if c.get("root") is not None else None,
               "bias": g.sum(0) if c.get("bias") is not None else None}
        if c.get("comp") is not None:
            basis = c["weight"].to(f64)
            out["comp"] = torch.einsum("rio,bio->rb", gw_eff, basis)
            out["weight"] = torch.einsum("rb,rio->bio", c["comp"].to(f64), gw_eff)
        else:
            out["weight"] = gw_eff
        gx = None
        if want_x:
            gagg = _mean_agg_transposed(g, edge_index, edge_type, n, r, cnt)
            wt = rb(torch.cat([w.transpose(1, 2).reshape(-1, d_in)] +
                              ([c["root"].to(f64).t()] if c.get("root") is not None else [])))   # one scale for [W ; root]
            k1 = r * w.size(2)
            gx = rb(gagg) @ wt[:k1]                   # the aggregate and g carry a scale each
            if c.get("root") is not None:
                gx = gx + rb(g) @ wt[k1:]
        return out, gx

    x = emb.to(f64)
    agg1, a1, z1 = layer_fwd(x, conv1)
    h = z1.clamp(min=0)
    agg2, a2, out = layer_fwd(h, conv2)
    g = cot.to(f64)
    g2, gh = layer_bwd(g, a2, conv2)
    mask = (h > 0) if relu_mask is None else relu_mask.to(torch.bool)
    gz = gh * mask
    g1, gx = layer_bwd(gz, a1, conv1)
    grads = {"emb": gx}
    for name, gd in (("conv1", g1), ("conv2", g2)):
        grads.update({f"{name}.{k}": v for k, v in gd.items() if v is not None})
    return {"out": out, "h": h, "grads": grads}


# --------------------------------------------------------------------------
# restatement #3 on a SAMPLE of rows (graphs too large to evaluate whole on the CPU: BASELINE configs[3])
# --------------------------------------------------------------------------
def encoder_rows_f64(emb, conv1: dict, conv2: dict, edge_index, edge_type, cot, out_rows, grad_rows,
                     relu_mask_fn):
    """``encoder_explicit_f64`` restricted to the neighbourhoods of a few rows: the same formulas
    (rgcn.py:117-130 forward; rows A3-A7 of SURVEY section 8a), float64, evaluated only where the
    requested rows depend on them.

    ``out_rows`` (LongTensor): rows of ``out = conv2(relu(conv1(emb)))`` wanted - needs ``h`` on their
    in-neighbours, i.e. every in-edge of ``out_rows`` and of those neighbours.
    ``grad_rows``: rows of ``d <out, cot> / d emb`` wanted - needs ``gz`` on their out-neighbours,
    i.e. every out-edge of ``grad_rows`` and of those neighbours; the mean divisors are the GLOBAL
    in-degree counts (integer work over the whole edge list).
    ``relu_mask_fn(nodes) -> bool [len(nodes), hidden]``: the ReLU decisions of those rows for the
    backward (the device's ``h > 0``, as in the full-size tests: a pre-activation within rounding of
    zero must not switch a unit's gradient on one side only).

    -> dict(out=[len(out_rows), d_out], grad_emb=[len(grad_rows), d_in], h_rows=(nodes, h values))"""
    f64 = torch.float64
    n = emb.size(0)
    r = conv1["weight"].size(0)
    src, dst = edge_index[0], edge_index[1]
    cnt = torch.bincount(dst * r + edge_type, minlength=n * r).clamp(min=1).to(f64)      # cnt[dst * R + rel]

    def member(nodes):
        lut = torch.zeros(n, dtype=torch.bool)
        lut[nodes] = True
        return lut

    def local(nodes):
        lut = torch.full((n,), -1, dtype=torch.int64)
        lut[nodes] = torch.arange(nodes.numel())
        return lut

    def wcat(c):
        w = c["weight"].to(f64)
        parts = [w.reshape(-1, w.size(-1))]
        if c.get("root") is not None:
            parts.append(c["root"].to(f64))
        return w, torch.cat(parts)

    def layer_rows(x_of, rows, c):
        """rows of one layer's pre-activation: x_of(nodes) -> float64 rows of the layer input"""
        m = member(rows)[dst]
        e_src, e_dst, e_rel = src[m], dst[m], edge_type[m]
        need = torch.unique(torch.cat([rows, e_src]))
        xin = x_of(need)
        lx = local(need)
        lr = local(rows)
        d = xin.size(1)
        key = lr[e_dst] * r + e_rel
        s = torch.zeros(rows.numel() * r, d, dtype=f64).index_add_(0, key, xin[lx[e_src]])
        agg = (s / cnt[(rows.view(-1, 1) * r + torch.arange(r)).view(-1)].view(-1, 1)).view(rows.numel(), r * d)
        a = torch.cat([agg, xin[lx[rows]]], 1) if c.get("root") is not None else agg
        z = a @ wcat(c)[1]
        if c.get("bias") is not None:
            z = z + c["bias"].to(f64)
        return z

    out = {}
    if out_rows is not None and out_rows.numel():
        rows0 = torch.unique(out_rows)
        s1 = torch.unique(torch.cat([rows0, src[member(rows0)[dst]]]))
        h1 = layer_rows(lambda nodes: emb[nodes].to(f64), s1, conv1).clamp(min=0)
        l1 = local(s1)
        z2 = layer_rows(lambda nodes: h1[l1[nodes]], rows0, conv2)
        out["out"] = z2[local(rows0)[out_rows]]
        out["h_rows"] = (s1, h1)

    def input_grad_rows(g_of, rows, c):
        """rows of one layer's input gradient from g_of(nodes) -> float64 rows of its output gradient"""
        w, _ = wcat(c)
        m = member(rows)[src]
        e_src, e_dst, e_rel = src[m], dst[m], edge_type[m]
        need = torch.unique(torch.cat([rows, e_dst]))
        gin = g_of(need)
        lg, lr = local(need), local(rows)
        d = gin.size(1)
        contrib = gin[lg[e_dst]] / cnt[e_dst * r + e_rel].view(-1, 1)
        gagg = torch.zeros(rows.numel() * r, d, dtype=f64).index_add_(0, lr[e_src] * r + e_rel, contrib)
        gx = gagg.view(rows.numel(), r * d) @ w.transpose(1, 2).reshape(r * d, -1)
        if c.get("root") is not None:
            gx = gx + gin[lg[rows]] @ c["root"].to(f64).t()
        return gx

    if grad_rows is not None and grad_rows.numel():
        k0 = torch.unique(grad_rows)
        t1 = torch.unique(torch.cat([k0, dst[member(k0)[src]]]))
        gz = input_grad_rows(lambda nodes: cot[nodes].to(f64), t1, conv2) * relu_mask_fn(t1).to(f64)
        lt = local(t1)
        gx = input_grad_rows(lambda nodes: gz[lt[nodes]], k0, conv1)
        out["grad_emb"] = gx[local(k0)[grad_rows]]
    return out

"""DistMult scoring head on the HIP library.

Mirror of the reference's ``LinkPredictor`` (``src/models/rgcn.py:145-243``): same
constructor ``(num_relations, embedding_dim, dropout=0.0)``, same parameter
(``relation_embeddings.weight`` [R, d], xavier-uniform), same methods
``forward(head_embeddings, tail_embeddings, relation_types)`` and
``score_all_tails(head_embeddings, relation_types, all_tail_embeddings)``.

Extra entry ``score_triples(node_embeddings, head_idx, tail_idx, relation_types)`` fuses
the two row gathers of ``rgcn.py:325-326`` into the scoring kernel (SURVEY row C1 + C2).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import ops


class _DistMultFunction(torch.autograd.Function):
    """scores[b] = sum_d H[hi(b)] * Rm[ri(b)] * T[ti(b)]; an index of None means row b."""

    @staticmethod
    def forward(ctx, h: Tensor, h_idx: Optional[Tensor], t: Tensor, t_idx: Optional[Tensor],
                r: Tensor, r_idx: Optional[Tensor]) -> Tensor:
        h, t, r = h.contiguous(), t.contiguous(), r.contiguous()
        h_idx = h_idx.contiguous() if h_idx is not None else None
        t_idx = t_idx.contiguous() if t_idx is not None else None
        r_idx = r_idx.contiguous() if r_idx is not None else None
        batch = (h_idx if h_idx is not None else h).size(0)
        scores = ops.distmult_fwd(h, h_idx, t, t_idx, r, r_idx, batch)
        ctx.batch = batch
        ctx.same_ht = h.data_ptr() == t.data_ptr() and h.shape == t.shape
        ctx.save_for_backward(h, h_idx, t, t_idx, r, r_idx)
        return scores

    @staticmethod
    def backward(ctx, gs: Tensor):
        h, h_idx, t, t_idx, r, r_idx = ctx.saved_tensors
        gs = gs.contiguous()
        need_h, _, need_t, _, need_r, _ = ctx.needs_input_grad

        def buf(src, idx, need):                     # (an indexed table is cleared by the backward's first launch)
            return torch.empty_like(src) if need else None

        gh = buf(h, h_idx, need_h)
        # head and tail gathered from one table: accumulate both into one buffer
        shared = ctx.same_ht and need_h and need_t and h_idx is not None and t_idx is not None
        gt = gh if shared else buf(t, t_idx, need_t)
        gr = buf(r, r_idx, need_r)
        ops.distmult_bwd(gs, h, h_idx, t, t_idx, r, r_idx, ctx.batch, gh, gt, gr, zero_tables=True)
        if shared:
            # autograd sums the two returned grads of the same leaf: hand back the whole
            # accumulation once and an untouched None for the second slot
            return gh, None, None, None, gr, None
        return gh, None, gt, None, gr, None


class _RelationRows(torch.autograd.Function):
    """``table[idx]`` for the [R, d] relation table.  Thousands of samples share R rows, which
    is the worst case for an atomic / sort based embedding backward (160 us per step measured
    with ``nn.Embedding``); here the table gradient is a fixed two-level tree over 256-sample
    segments (``ops.segment_sum``) -- deterministic, a few microseconds."""

    @staticmethod
    def forward(ctx, table: Tensor, idx: Tensor) -> Tensor:
        ctx.save_for_backward(idx)
        ctx.rows = table.size(0)
        return table.index_select(0, idx)

    @staticmethod
    def backward(ctx, g: Tensor):
        (idx,) = ctx.saved_tensors
        return ops.segment_sum(g.contiguous(), idx.contiguous(), ctx.rows), None


class _DistMultBCEFunction(torch.autograd.Function):
    """(mean binary_cross_entropy_with_logits(scores, labels), scores): the scoring kernel also
    emits the per-sample loss, and ONE backward kernel turns the gradient of the mean loss into
    the head / tail / relation gradients (sigmoid, subtraction, 1/B and the DistMult products
    fused) - instead of the eight elementwise launches of BCEWithLogitsLoss around the head
    (``src/train.py:300``).  ``scores`` is returned for the accuracy bookkeeping only."""

    @staticmethod
    def forward(ctx, h, h_idx, t, t_idx, r, r_idx, labels, stats=None):
        h, t, r = h.contiguous(), t.contiguous(), r.contiguous()
        h_idx, t_idx, r_idx = (i.contiguous() if i is not None else None for i in (h_idx, t_idx, r_idx))
        labels = labels.contiguous()
        batch = (h_idx if h_idx is not None else h).size(0)
        scores, per_sample = ops.distmult_bce_fwd(h, h_idx, t, t_idx, r, r_idx, labels, batch)
        ctx.batch = batch
        ctx.same_ht = h.data_ptr() == t.data_ptr() and h.shape == t.shape
        ctx.save_for_backward(h, h_idx, t, t_idx, r, r_idx, labels, scores)
        ctx.mark_non_differentiable(scores)
        ctx.set_materialize_grads(False)             # (no zero-filled stand-in for the gradient of `scores` per step)
        # the mean in a fixed order and, when the caller keeps running sums on the device (``stats`` = (loss_sum,
        # correct, cursor, cursor_add), any of the first three None), the step's bookkeeping in the same launch
        loss_sum, correct, cursor, cursor_add = stats if stats is not None else (None, None, None, 0)
        mean = ops.distmult_bce_reduce(per_sample, scores, labels, loss_sum, correct, cursor, cursor_add)
        return mean.reshape(()), scores

    @staticmethod
    def backward(ctx, g_loss, _g_scores):
        h, h_idx, t, t_idx, r, r_idx, labels, scores = ctx.saved_tensors
        need_h, _, need_t, _, need_r, _, _, _ = ctx.needs_input_grad
        if g_loss is None:                           # nothing downstream used the loss
            return (None,) * 8

        def buf(src, idx, need):                     # (an indexed table is cleared by the backward's first launch)
            return torch.empty_like(src) if need else None

        gh = buf(h, h_idx, need_h)
        shared = ctx.same_ht and need_h and need_t and h_idx is not None and t_idx is not None
        gt = gh if shared else buf(t, t_idx, need_t)
        gr = buf(r, r_idx, need_r)
        ops.distmult_bce_bwd(g_loss.reshape(1).contiguous(), scores, labels, h, h_idx, t, t_idx, r, r_idx,
                             ctx.batch, gh, gt, gr, zero_tables=True)
        if shared:
            return gh, None, None, None, gr, None, None, None
        return gh, None, gt, None, gr, None, None, None


class _ScoreAllTailsFunction(torch.autograd.Function):
    """``(head * rel[rel_idx]) @ emb.T``; the gradients are three small dense products (the reference never asks for
    them - ``score_all_tails`` runs under ``no_grad`` in ``evaluate.py`` - so they stay on the library GEMM)."""

    @staticmethod
    def forward(ctx, head: Tensor, rel: Tensor, rel_idx: Tensor, emb: Tensor) -> Tensor:
        head, rel, emb = head.contiguous(), rel.contiguous(), emb.contiguous()
        rel_idx = rel_idx.contiguous()
        scores, hr = ops.distmult_score_all_tails(head, rel, rel_idx, emb)
        ctx.save_for_backward(head, rel, rel_idx, emb, hr)
        return scores

    @staticmethod
    def backward(ctx, g: Tensor):
        head, rel, rel_idx, emb, hr = ctx.saved_tensors
        g = g.contiguous()
        grad_head = grad_rel = grad_emb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_hr = g @ emb
            if ctx.needs_input_grad[0]:
                grad_head = grad_hr * rel[rel_idx]
            if ctx.needs_input_grad[1]:
                grad_rel = torch.zeros_like(rel).index_add_(0, rel_idx, grad_hr * head)
        if ctx.needs_input_grad[3]:
            grad_emb = g.t() @ hr
        return grad_head, grad_rel, None, grad_emb


def distmult(h, h_idx, t, t_idx, r, r_idx) -> Tensor:
    return _DistMultFunction.apply(h, h_idx, t, t_idx, r, r_idx)


class LinkPredictor(nn.Module):
    """DistMult decoder: ``score(h, r, t) = sum_d h_d * r_d * t_d``."""

    def __init__(self, num_relations: int, embedding_dim: int, dropout: float = 0.0):
        super().__init__()
        self.num_relations = num_relations
        self.embedding_dim = embedding_dim
        self.relation_embeddings = nn.Embedding(num_relations, embedding_dim)
        self.dropout = nn.Dropout(dropout)
        self._init_embeddings()

    def _init_embeddings(self) -> None:
        nn.init.xavier_uniform_(self.relation_embeddings.weight)

    def _relation_operand(self, relation_types: Tensor):
        """(matrix, index) for the relation factor.  With active dropout the reference
        drops the *gathered* [B, d] rows (rgcn.py:207-208), one mask per sample, so the
        gather + dropout stay torch ops and the kernel reads the dropped rows; otherwise the
        kernel gathers straight from the [R, d] table."""
        if self.training and self.dropout.p > 0:
            return self.dropout(_RelationRows.apply(self.relation_embeddings.weight, relation_types)), None
        return self.relation_embeddings.weight, relation_types

    def forward(self, head_embeddings: Tensor, tail_embeddings: Tensor,
                relation_types: Tensor) -> Tensor:
        r, r_idx = self._relation_operand(relation_types)
        return distmult(head_embeddings, None, tail_embeddings, None, r, r_idx)

    def score_triples(self, node_embeddings: Tensor, head_indices: Tensor, tail_indices: Tensor,
                      relation_types: Tensor) -> Tensor:
        """``forward(node_embeddings[head], node_embeddings[tail], rel)`` without
        materialising the two [B, d] gathers."""
        r, r_idx = self._relation_operand(relation_types)
        return distmult(node_embeddings, head_indices, node_embeddings, tail_indices, r, r_idx)

    def bce_loss(self, node_embeddings: Tensor, head_indices: Tensor, tail_indices: Tensor,
                 relation_types: Tensor, labels: Tensor, stats=None):
        """``(BCEWithLogitsLoss()(score_triples(...), labels), scores)`` as one fused node.  ``stats``: the caller's
        device-resident running sums and batch cursor ``(loss_sum, correct, cursor, cursor_add)``, updated by the launch
        that forms the mean (``Trainer``: ``src/train.py:321-326`` without nine elementwise launches per step)."""
        r, r_idx = self._relation_operand(relation_types)
        return _DistMultBCEFunction.apply(node_embeddings, head_indices, node_embeddings, tail_indices,
                                          r, r_idx, labels, stats)

    @torch.no_grad()
    def rank_tails(self, head_embeddings: Tensor, relation_types: Tensor, all_tail_embeddings: Tensor,
                   tail_indices: Tensor) -> Tensor:
        """1-based rank of ``tail_indices[b]`` among all entities for ``(head, relation)``:
        ``argsort(score_all_tails(...)[b], descending=True)`` position + 1, as
        ``evaluate.py:266-274`` computes it, but from one fused MFMA pass that counts the
        candidates beating the true tail's score (ties, measure zero in fp32, rank first)."""
        hr = (head_embeddings * self.relation_embeddings(relation_types)).contiguous()
        emb = all_tail_embeddings.contiguous()
        true_score = (hr * emb[tail_indices]).sum(1)
        return ops.distmult_rank_tails(hr, emb, true_score, tail_indices.contiguous())

    def score_all_tails(self, head_embeddings: Tensor, relation_types: Tensor,
                        all_tail_embeddings: Tensor) -> Tensor:
        """``(h * r) @ E^T`` -> [B, num_entities] (rgcn.py:215-243): the ranking kernel's GEMM with a store
        epilogue (``distmult_score_all_tails``), so ``score_all_tails(...)[b, n]`` and the score ``rank_tails``
        compares are the same bits.  Differentiable like the reference's expression.  An embedding width that is not a
        multiple of the kernel's 32-wide k step (the reference's is 128) takes the library GEMM on the GPU."""
        if head_embeddings.size(1) % 32 and head_embeddings.is_cuda:
            return (head_embeddings * self.relation_embeddings(relation_types)) @ all_tail_embeddings.t()
        return _ScoreAllTailsFunction.apply(head_embeddings, self.relation_embeddings.weight, relation_types,
                                            all_tail_embeddings)

"""Embedding consumers (SURVEY.md section 8f "next" row 4).

The reference's analysis scripts use the encoder output only through cosine similarity
mapped to [0, 1]:

* ``compare_methods.RGCNMethod.predict`` (``src/compare_methods.py:368-382``): per-pair
  ``(cos(drug, disease) + 1) / 2``;
* ``compare_methods.RGCNMethod.predict_all`` (``compare_methods.py:384-397``): the
  ``[n_drug, n_disease]`` matrix of the same;
* ``case_studies.predict_top_drugs`` (``src/case_studies.py:236-284``): the top-k drugs of one
  disease above a threshold, best first.

Here they stay on the device: rows are normalised once, the per-pair form reuses the DistMult
kernel (relation factor = ones, so ``sum_d h*1*t`` of unit rows is the cosine), the matrix
form is a plain library GEMM, the ranking a stable descending sort (ties keep candidate
order, as the reference's ``list.sort(reverse=True)`` does).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import ops

_Index = Union[Tensor, Sequence[int]]


def _index(idx: _Index, device) -> Tensor:
    t = idx if isinstance(idx, Tensor) else torch.as_tensor(list(idx), dtype=torch.int64)
    return t.to(device=device, dtype=torch.int64).contiguous()


def normalize_rows(embeddings: Tensor) -> Tensor:
    """``emb / ||emb||`` per row, as ``compare_methods.py:390-391`` (no epsilon: a zero row
    gives nan there too)."""
    if embeddings.dim() != 2:
        raise ValueError("embeddings must be [N, d]")
    return (embeddings / embeddings.norm(dim=1, keepdim=True)).contiguous()


@torch.no_grad()
def cosine_pair_scores(embeddings: Tensor, drug_indices: _Index, disease_indices: _Index,
                       normalized: bool = False) -> Tensor:
    """``(cos(emb[drug_b], emb[disease_b]) + 1) / 2`` for each pair b."""
    unit = embeddings if normalized else normalize_rows(embeddings)
    a, b = _index(drug_indices, unit.device), _index(disease_indices, unit.device)
    if a.shape != b.shape or a.dim() != 1:
        raise ValueError("drug_indices and disease_indices must be 1-D and equally long")
    ones = torch.ones(1, unit.size(1), device=unit.device, dtype=unit.dtype)
    zero = torch.zeros(a.numel(), dtype=torch.int64, device=unit.device)
    cos = ops.distmult_fwd(unit, a, unit, b, ones, zero, a.numel())
    return (cos + 1) / 2


@torch.no_grad()
def cosine_score_matrix(embeddings: Tensor, drug_indices: _Index, disease_indices: _Index,
                        normalized: bool = False) -> Tensor:
    """``[n_drug, n_disease]`` matrix of ``(cos + 1) / 2``."""
    unit = embeddings if normalized else normalize_rows(embeddings)
    a, b = _index(drug_indices, unit.device), _index(disease_indices, unit.device)
    return (unit.index_select(0, a) @ unit.index_select(0, b).t() + 1) / 2


@torch.no_grad()
def predict_top_drugs(embeddings: Tensor, disease_idx: int, drug_indices: _Index, top_k: int = 10,
                      threshold: float = 0.0, normalized: bool = False) -> List[Tuple[int, float]]:
    """[(drug_idx, score)] of the ``top_k`` drugs with score >= threshold for one disease,
    best first."""
    unit = embeddings if normalized else normalize_rows(embeddings)
    cand = _index(drug_indices, unit.device)
    scores = (torch.mv(unit.index_select(0, cand), unit[disease_idx]) + 1) / 2
    keep = scores >= threshold
    cand, scores = cand[keep], scores[keep]
    order = torch.sort(scores, descending=True, stable=True).indices[:top_k]
    return list(zip(cand[order].tolist(), scores[order].tolist()))

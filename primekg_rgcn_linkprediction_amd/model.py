"""Encoder + full link-prediction model on the HIP R-GCN engine.

Host-side mirror of the reference's model file so that its callers keep working
unchanged: class names, constructor arguments and order,
method names/arguments and state-dict keys follow ``src/models/rgcn.py``:

* ``DrugDiseaseRGCN``  - ``rgcn.py:21-142``  (embedding table -> conv1 -> relu -> dropout -> conv2)
* ``DrugDiseaseModel`` - ``rgcn.py:246-415`` (encoder + DistMult decoder, ``forward`` /
  ``predict`` / ``predict_all_tails`` / ``get_embeddings``)

so ``Trainer`` (``src/train.py:291-297, 389-395``) and ``ModelEvaluator``
(``src/evaluate.py:189-195, 251-262``) call sites stay textually intact, and reference
checkpoints (``encoder.node_embeddings.weight``, ``encoder.conv{1,2}.{weight,root,bias[,comp]}``,
``decoder.relation_embeddings.weight``) load with ``load_state_dict``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .conv import RGCNConv, rgcn_encoder2
from .head import LinkPredictor


class DrugDiseaseRGCN(nn.Module):
    """Two-layer R-GCN encoder over a learnable node-embedding table."""

    def __init__(self, num_nodes: int, num_relations: int, embedding_dim: int = 64,
                 hidden_dim: int = 128, dropout: float = 0.5, num_bases: Optional[int] = None,
                 gather_dtype=None):
        super().__init__()
        self.num_nodes = num_nodes
        self.num_relations = num_relations
        self.embedding_dim = embedding_dim
        self.hidden_dim = hidden_dim
        self.node_embeddings = nn.Embedding(num_nodes, embedding_dim)
        self.conv1 = RGCNConv(in_channels=embedding_dim, out_channels=hidden_dim,
                              num_relations=num_relations, num_bases=num_bases, gather_dtype=gather_dtype)
        self.conv2 = RGCNConv(in_channels=hidden_dim, out_channels=hidden_dim,
                              num_relations=num_relations, num_bases=num_bases, gather_dtype=gather_dtype)
        self.dropout = nn.Dropout(dropout)
        self._init_embeddings()

    def _init_embeddings(self) -> None:
        nn.init.xavier_uniform_(self.node_embeddings.weight)

    def forward(self, edge_index: Tensor, edge_type: Tensor,
                node_indices: Optional[Tensor] = None) -> Tensor:
        # the whole table is the layer-1 input (no lookup) unless a subset is asked for
        x = self.node_embeddings.weight if node_indices is None else self.node_embeddings(node_indices)
        p = self.dropout.p if self.training else 0.0
        if p < 1.0:
            # conv1 -> relu -> dropout -> conv2 (rgcn.py:123-128) as one fused autograd node; the
            # mask comes from torch's dropout kernel and RNG stream, as in the reference
            return rgcn_encoder2(x, edge_index, edge_type, self.conv1, self.conv2, dropout_p=p)
        x = self.conv1(x, edge_index, edge_type, activation="relu")      # p == 1: everything dropped
        x = self.dropout(x)
        return self.conv2(x, edge_index, edge_type)

    def get_node_embeddings(self, node_indices: Tensor) -> Tensor:
        return self.node_embeddings(node_indices)


class DrugDiseaseModel(nn.Module):
    """R-GCN encoder + DistMult decoder."""

    def __init__(self, num_nodes: int, num_relations: int, embedding_dim: int = 64,
                 hidden_dim: int = 128, dropout: float = 0.5, decoder_dropout: float = 0.0,
                 num_bases: Optional[int] = None, gather_dtype=None):
        super().__init__()
        self.num_nodes = num_nodes
        self.num_relations = num_relations
        self.hidden_dim = hidden_dim
        self.encoder = DrugDiseaseRGCN(num_nodes=num_nodes, num_relations=num_relations,
                                       embedding_dim=embedding_dim, hidden_dim=hidden_dim,
                                       dropout=dropout, num_bases=num_bases, gather_dtype=gather_dtype)
        self.decoder = LinkPredictor(num_relations=num_relations, embedding_dim=hidden_dim,
                                     dropout=decoder_dropout)

    def forward(self, edge_index: Tensor, edge_type: Tensor, head_indices: Tensor,
                tail_indices: Tensor, relation_types: Tensor) -> Tensor:
        node_embeddings = self.encoder(edge_index, edge_type)
        # rgcn.py:325-329: two row gathers + decoder, fused into one scoring kernel
        return self.decoder.score_triples(node_embeddings, head_indices, tail_indices, relation_types)

    def bce_loss(self, edge_index: Tensor, edge_type: Tensor, head_indices: Tensor, tail_indices: Tensor,
                 relation_types: Tensor, labels: Tensor, stats=None):
        """``(nn.BCEWithLogitsLoss()(self(...), labels), scores)`` - what ``Trainer`` needs per
        step (``train.py:291-300``) with the loss fused into the head kernels; ``stats``: see
        ``LinkPredictor.bce_loss``."""
        node_embeddings = self.encoder(edge_index, edge_type)
        return self.decoder.bce_loss(node_embeddings, head_indices, tail_indices, relation_types, labels, stats)

    def predict(self, edge_index: Tensor, edge_type: Tensor, head_indices: Tensor,
                tail_indices: Tensor, relation_types: Tensor) -> Tensor:
        self.eval()
        with torch.no_grad():
            return self.forward(edge_index, edge_type, head_indices, tail_indices, relation_types)

    def predict_all_tails(self, edge_index: Tensor, edge_type: Tensor, head_indices: Tensor,
                          relation_types: Tensor) -> Tensor:
        self.eval()
        with torch.no_grad():
            emb = self.encoder(edge_index, edge_type)
            return self.decoder.score_all_tails(emb[head_indices], relation_types, emb)

    def get_embeddings(self, edge_index: Tensor, edge_type: Tensor) -> Tensor:
        self.eval()
        with torch.no_grad():
            return self.encoder(edge_index, edge_type)

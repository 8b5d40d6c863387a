"""Evaluation protocol on the HIP path (SURVEY.md section 8f "next" row 2).

Counterpart of the metric core of the reference's ``src/evaluate.py`` (plots, reports and the
CLI stay out of scope): same protocol, same metric names and return shapes, with the two
hot spots removed -

* ``compute_scores_and_labels`` (``evaluate.py:147-217``): positives + random corruptions of the
  test triples scored over the FULL graph.  The encoder runs ONCE, not once per 1,024-edge batch.
* ``compute_ranking_metrics`` (``evaluate.py:219-299``): rank of the true tail among all
  entities.  The reference re-encodes per batch and, per test edge, argsorts 30,926 scores in a
  Python loop; here the encoder runs once and ``LinkPredictor.rank_tails`` returns every rank
  from one fused MFMA pass (no [B, N] score matrix, no sort).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .model import DrugDiseaseModel
from .train import NegativeSampler


class ModelEvaluator:
    def __init__(self, model: DrugDiseaseModel, test_data: Dict, full_graph: Dict, device: torch.device,
                 batch_size: int = 1024):
        self.model = model.to(device).eval()
        self.device, self.batch_size = device, batch_size
        self.test_edge_index = test_data["edge_index"].to(device)
        self.test_edge_type = test_data["edge_type"].to(device)
        self.full_edge_index = full_graph["edge_index"].to(device)
        self.full_edge_type = full_graph["edge_type"].to(device)
        self.num_nodes = int(full_graph["num_nodes"])
        self.num_test_edges = int(self.test_edge_index.size(1))
        self._emb = None

    @torch.no_grad()
    def embeddings(self) -> torch.Tensor:
        """node embeddings of the full graph, encoded once and kept"""
        if self._emb is None:
            self._emb = self.model.encoder(self.full_edge_index, self.full_edge_type)
        return self._emb

    @torch.no_grad()
    def compute_scores_and_labels(self, num_neg_samples: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """-> (sigmoid scores, labels): every test column + ``num_neg_samples`` corruptions."""
        emb = self.embeddings()
        sampler = NegativeSampler(self.num_nodes, num_neg_samples)
        head, tail, rel = self.test_edge_index[0], self.test_edge_index[1], self.test_edge_type
        scores, labels = [], []
        for lo in range(0, self.num_test_edges, self.batch_size):
            h, t, r = head[lo: lo + self.batch_size], tail[lo: lo + self.batch_size], rel[lo: lo + self.batch_size]
            nh, nt, nr = sampler.sample(h, t, r)
            s = self.model.decoder.score_triples(emb, torch.cat([h, nh]), torch.cat([t, nt]), torch.cat([r, nr]))
            scores.append(torch.sigmoid(s))
            labels.append(torch.cat([torch.ones(h.numel(), device=self.device),
                                     torch.zeros(nh.numel(), device=self.device)]))
        return torch.cat(scores).cpu().numpy(), torch.cat(labels).cpu().numpy()

    @torch.no_grad()
    def tail_ranks(self) -> torch.Tensor:
        """int64 [num_test_edges]: 1-based rank of every true tail among all entities"""
        emb = self.embeddings()
        head, tail, rel = self.test_edge_index[0], self.test_edge_index[1], self.test_edge_type
        return self.model.decoder.rank_tails(emb[head], rel, emb, tail)

    def compute_ranking_metrics(self, k_values: Sequence[int] = (10, 50)) -> Dict:
        ranks = self.tail_ranks().cpu().numpy().astype(np.float64)
        metrics = {"mrr": float(np.mean(1.0 / ranks)), "mean_rank": float(np.mean(ranks)),
                   "median_rank": float(np.median(ranks))}
        for k in k_values:
            metrics[f"hits@{k}"] = float(np.mean(ranks <= k))
        return metrics

    @staticmethod
    def compute_classification_metrics(scores: np.ndarray, labels: np.ndarray, threshold: float = 0.5) -> Dict:
        from sklearn.metrics import (average_precision_score, f1_score, precision_score, recall_score,
                                     roc_auc_score)
        pred = (scores >= threshold).astype(int)
        return {"auc_roc": float(roc_auc_score(labels, scores)),
                "auc_pr": float(average_precision_score(labels, scores)),
                "precision": float(precision_score(labels, pred)), "recall": float(recall_score(labels, pred)),
                "f1_score": float(f1_score(labels, pred)), "threshold": threshold}

    def evaluate(self, num_neg_samples: int = 1, k_values: List[int] = (10, 50)) -> Dict:
        scores, labels = self.compute_scores_and_labels(num_neg_samples)
        return {"classification": self.compute_classification_metrics(scores, labels),
                "ranking": self.compute_ranking_metrics(k_values),
                "num_test_edges": self.num_test_edges, "num_nodes": self.num_nodes}

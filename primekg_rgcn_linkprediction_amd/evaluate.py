"""Evaluation protocol on the HIP path (SURVEY.md section 8f "next" row 2).

Counterpart of the reference's ``src/evaluate.py`` minus its plots: same command line
(``--model_path --data_dir --output_dir --batch_size --num_neg_samples --k_values --device``,
``evaluate.py:766-827``), same checkpoint reading (N and R recovered from the state dict,
``evaluate.py:655-730``), same protocol, metric names and return shapes, same ``results.json`` /
``metrics_summary.txt`` (``evaluate.py:594-652``; cf. ``results_final/results.json``), with the two
hot spots removed -

* ``compute_scores_and_labels`` (``evaluate.py:147-217``): positives + random corruptions of the
  test triples scored over the FULL graph.  The encoder runs ONCE, not once per 1,024-edge batch.
* ``compute_ranking_metrics`` (``evaluate.py:219-299``): rank of the true tail among all
  entities.  The reference re-encodes per batch and, per test edge, argsorts 30,926 scores in a
  Python loop; here the encoder runs once and ``LinkPredictor.rank_tails`` returns every rank
  from one fused MFMA pass (no [B, N] score matrix, no sort).
"""
from __future__ import annotations

import argparse
import json
import logging
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .model import DrugDiseaseModel
from .train import NegativeSampler, filter_edges

logger = logging.getLogger("primekg_rgcn_linkprediction_amd.evaluate")


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
        self.scores, self.labels = scores, labels          # kept for plotting code, as the reference does
        metrics = {"classification": self.compute_classification_metrics(scores, labels),
                   "ranking": self.compute_ranking_metrics(k_values),
                   "test_edges": self.num_test_edges, "num_nodes": self.num_nodes}
        from . import ops
        ops.check_indices(self.device)                     # an id outside the embedding table anywhere above: IndexError
        return metrics


# ------------------------------------------------------------------------------------------
# checkpoint / data / results files / CLI
# ------------------------------------------------------------------------------------------
def load_model(model_path: str, device: torch.device, trust_pickle: bool = False) -> Tuple[DrugDiseaseModel, Dict]:
    """-> (model in eval mode on ``device``, model_info).  A training checkpoint pickles its
    argparse ``Namespace`` next to the tensors (``train.py:431-442``; the reference therefore loads with
    ``weights_only=False``, ``evaluate.py:672``).  Here the file is read with the restricted unpickler
    (``weights_only=True``) that is allowed exactly one extra class, ``argparse.Namespace`` - nothing in the
    file can execute.  ``trust_pickle=True`` (CLI ``--trust_checkpoint``) falls back to the unrestricted
    loader for checkpoints that hold other objects: only for files you wrote yourself."""
    import argparse
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
    except Exception as exc:
        if not trust_pickle:
            raise RuntimeError(f"{model_path} holds objects the restricted loader refuses ({exc}); re-run with "
                               f"--trust_checkpoint if (and only if) you wrote this file yourself") from exc
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
    args = checkpoint.get("args")
    if args is None:
        raise ValueError("Checkpoint does not contain 'args'. Cannot reconstruct model architecture.")
    state = checkpoint["model_state_dict"]
    num_nodes = state["encoder.node_embeddings.weight"].size(0)
    num_relations = state["decoder.relation_embeddings.weight"].size(0)
    model = DrugDiseaseModel(num_nodes=num_nodes, num_relations=num_relations, embedding_dim=args.embedding_dim,
                             hidden_dim=args.hidden_dim, dropout=args.dropout,
                             decoder_dropout=getattr(args, "decoder_dropout", 0.0),
                             num_bases=getattr(args, "num_bases", None))
    model.load_state_dict(state)
    model = model.to(device).eval()
    info = {"checkpoint_path": str(model_path), "epoch": checkpoint.get("epoch", "unknown"), "num_nodes": num_nodes,
            "num_relations": num_relations, "embedding_dim": args.embedding_dim, "hidden_dim": args.hidden_dim,
            "num_parameters": sum(p.numel() for p in model.parameters())}
    for key in ("best_val_loss", "best_val_acc"):
        if key in checkpoint:
            info[key] = checkpoint[key]
    return model, info


def load_test_data(data_dir: str) -> Tuple[Dict, Dict]:
    """``test_data.pt`` and ``full_graph.pt`` (tensor-only dicts: ``weights_only=True``) with the
    reference's out-of-range filter applied."""
    root = Path(data_dir)
    test = torch.load(root / "test_data.pt", weights_only=True)
    full = torch.load(root / "full_graph.pt", weights_only=True)
    n = test["num_nodes"]
    return filter_edges(test, n, "Test"), filter_edges(full, n, "Full graph")


def save_results(metrics: Dict, output_dir: Path, model_info: Optional[Dict] = None) -> None:
    """``results.json`` ({"metrics", "model_info"}) and ``metrics_summary.txt`` in the reference's layout."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    with open(output_dir / "results.json", "w") as fh:
        json.dump({"metrics": metrics, "model_info": model_info or {}}, fh, indent=2)
    rule = "=" * 60
    lines = [rule, "EVALUATION RESULTS SUMMARY", rule, ""]
    if model_info:
        lines += ["Model Information:", "-" * 60] + [f"{k}: {v}" for k, v in model_info.items()] + [""]
    lines += ["Dataset Statistics:", "-" * 60, f"Test edges: {metrics['test_edges']:,}",
              f"Number of nodes: {metrics['num_nodes']:,}", ""]
    for title, key in (("Classification Metrics:", "classification"), ("Ranking Metrics:", "ranking")):
        lines += [title, "-" * 60] + [f"{k}: {v:.4f}" for k, v in metrics[key].items()] + [""]
    lines.append(rule)
    (output_dir / "metrics_summary.txt").write_text("\n".join(lines) + "\n")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Evaluate a trained R-GCN link predictor on MI355X")
    p.add_argument("--model_path", type=str, required=True)
    p.add_argument("--data_dir", type=str, default="data/processed")
    p.add_argument("--output_dir", type=str, default="results")
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--num_neg_samples", type=int, default=1)
    p.add_argument("--k_values", type=int, nargs="+", default=[10, 50])
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--trust_checkpoint", action="store_true",
                   help="allow the unrestricted pickle loader for --model_path (only for files you wrote yourself)")
    return p


def main(argv=None) -> Dict:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    args = build_parser().parse_args(argv)
    device = torch.device(args.device)
    model, info = load_model(args.model_path, device, trust_pickle=args.trust_checkpoint)
    test_data, full_graph = load_test_data(args.data_dir)
    evaluator = ModelEvaluator(model, test_data, full_graph, device, batch_size=args.batch_size)
    metrics = evaluator.evaluate(num_neg_samples=args.num_neg_samples, k_values=args.k_values)
    save_results(metrics, Path(args.output_dir), info)
    logger.info("Results saved to: %s", args.output_dir)
    return metrics


if __name__ == "__main__":
    main()

"""Training harness around the HIP path (SURVEY.md section 8f "next" row 1).

Counterpart of the reference's ``src/train.py``: same command-line flags and defaults
(``train.py:635-770``), same on-disk dict format (``train.py:563-567``) and host-side range
filter (``train.py:571-586``), same sampler semantics (``train.py:59-97``), same step
(positives + negatives -> ``model(train graph, heads, tails, rels)`` -> BCE-with-logits ->
backward -> clip-grad-norm -> Adam; ``train.py:276-318``) and the same validation protocol on the
full graph (``train.py:389-395``), so a reference user can run

    python -m primekg_rgcn_linkprediction_amd.train --data_dir data/processed --epochs 100

and get checkpoints with the reference's keys (``train.py:431-442``).  What differs, on
purpose: the model is this package's ``DrugDiseaseModel`` (HIP kernels); the running loss /
accuracy are accumulated on the device and read back once per epoch instead of two ``.item()``
host syncs per step (``train.py:322,325``); a mini-batch (slice of the shuffled columns,
negatives, labels) is assembled by one kernel (``--torch_sampler`` restores the reference's
``torch.rand`` / ``torch.randint`` sequence); BCE-with-logits is fused into the head kernels;
gradient clipping + the Adam/AdamW update are two launches on the optimizer's own state
(``--torch_optimizer`` keeps torch's kernels); and the whole step is one captured HIP graph
replayed per batch (``--no_hip_graph`` launches it eagerly).  Same update rule throughout.

``--synthetic`` builds a PrimeKG-shaped random graph instead of loading ``--data_dir`` (the
reference's ``train_data.pt`` / ``full_graph.pt`` blobs are not in its mount).
"""
from __future__ import annotations

import argparse
import logging
import time
from pathlib import Path
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .model import DrugDiseaseModel

logger = logging.getLogger("primekg_rgcn_linkprediction_amd.train")


class NegativeSampler:
    """Corrupt the head or the tail (fair coin per sample) of each positive triple with a
    uniformly random node.  Draw order matches the reference (one ``torch.rand`` for the coin,
    then one ``torch.randint`` for the replacement entity), so a seeded run consumes the RNG
    stream identically (fixture ``tests/golden/ref_negative_sampler.npz``)."""

    def __init__(self, num_nodes: int, num_neg_samples: int = 1):
        self.num_nodes = num_nodes
        self.num_neg_samples = num_neg_samples

    def sample(self, pos_head: torch.Tensor, pos_tail: torch.Tensor,
               pos_rel: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        k = self.num_neg_samples
        head, tail, rel = (t.repeat_interleave(k) for t in (pos_head, pos_tail, pos_rel))
        total = head.numel()
        flip_head = torch.rand(total, device=head.device) < 0.5
        entity = torch.randint(0, self.num_nodes, (total,), device=head.device)
        return torch.where(flip_head, entity, head), torch.where(flip_head, tail, entity), rel


def group_has_extras(optimizer: torch.optim.Optimizer) -> bool:
    """options of torch's Adam the two-launch update does not implement"""
    g = optimizer.param_groups
    return len(g) != 1 or bool(g[0].get("amsgrad")) or bool(g[0].get("maximize")) or len(g[0]["params"]) > 32


class Trainer:
    """Epoch loop, validation and checkpoints; attribute names follow the reference's
    ``Trainer`` so that scripts poking at ``trainer.train_losses`` etc. keep working."""

    def __init__(self, model: DrugDiseaseModel, train_data: Dict, val_data: Dict, full_graph: Dict,
                 device: torch.device, args: argparse.Namespace):
        self.model = model.to(device)
        self.device, self.args = device, args
        self.train_data, self.val_data, self.full_graph = train_data, val_data, full_graph
        # the three graphs live on the device for the whole run; each is bucketed once
        self.train_edge_index = train_data["edge_index"].to(device)
        self.train_edge_type = train_data["edge_type"].to(device)
        self.val_edge_index = val_data["edge_index"].to(device)
        self.val_edge_type = val_data["edge_type"].to(device)
        self.full_edge_index = full_graph["edge_index"].to(device)
        self.full_edge_type = full_graph["edge_type"].to(device)
        if device.type == "cuda":      # bucket now (or import the persisted structure, --bucket_cache)
            from . import ops
            for d, ei, et in ((train_data, self.train_edge_index, self.train_edge_type),
                              (full_graph, self.full_edge_index, self.full_edge_type)):
                ops.bucket(ei, et, d["num_nodes"], d["num_relations"], sidecar=d.get("sidecar"))
        opt = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW}.get(args.optimizer)
        if opt is None:
            raise ValueError(f"Unknown optimizer: {args.optimizer}")
        # whole-step HIP graph (sampler -> encoder -> head -> BCE -> backward -> clip -> Adam ->
        # running metrics) for the full-size batches of an epoch; needs a device-side Adam step count
        self.use_hip_graph = (device.type == "cuda" and not getattr(args, "no_hip_graph", False)
                              and max(1, getattr(args, "gradient_accumulation_steps", 1)) == 1)
        # on the GPU: the single-kernel update (same rule; the default one runs ~10 list kernels
        # per step, ~125 us here), with its step count on the device when the step is captured
        extra = {"fused": True, "capturable": self.use_hip_graph} if device.type == "cuda" else {}
        self.optimizer = opt(self.model.parameters(), lr=args.lr, weight_decay=args.weight_decay, **extra)
        self._graph = self._loss_sum = None
        self._grad_seed = {}
        # clip + Adam(W) as two launches of this package instead of ~14 of torch's
        self.fused_update = (device.type == "cuda" and not getattr(args, "torch_optimizer", False)
                             and not group_has_extras(self.optimizer))
        # mini-batch assembly (slice + negatives + labels) as one kernel; --torch_sampler keeps the
        # reference's op-by-op sampler on torch's RNG stream
        self.device_sampler = device.type == "cuda" and not getattr(args, "torch_sampler", False)
        self.criterion = nn.BCEWithLogitsLoss()
        self.neg_sampler = NegativeSampler(train_data["num_nodes"], args.num_neg_samples)
        self.best_val_loss, self.best_val_acc = float("inf"), 0.0
        self.train_losses, self.val_losses, self.train_accs, self.val_accs = [], [], [], []
        self.output_dir = Path(args.output_dir)
        self.checkpoint_dir, self.model_dir = self.output_dir / "checkpoints", self.output_dir / "models"
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        self.model_dir.mkdir(parents=True, exist_ok=True)

    @property
    def _fused_bookkeeping(self) -> bool:
        """the criterion, the running loss / hit sums and the batch cursor inside the head's launches"""
        return (self.device.type == "cuda" and isinstance(self.criterion, nn.BCEWithLogitsLoss)
                and hasattr(self.model, "bce_loss"))

    # -- batches ------------------------------------------------------------------------
    def _batches(self, edge_index, edge_type, shuffle: bool):
        """Yield (head, tail, rel) column slices; the permutation is drawn on the host like the
        reference's ``torch.randperm(num_edges)`` and moved to the device once."""
        e = edge_index.size(1)
        order = (torch.randperm(e) if shuffle else torch.arange(e)).to(edge_index.device)
        for lo in range(0, e, self.args.batch_size):
            idx = order[lo: lo + self.args.batch_size]
            yield edge_index[0, idx], edge_index[1, idx], edge_type[idx]

    def _with_negatives(self, head, tail, rel):
        nh, nt, nr = self.neg_sampler.sample(head, tail, rel)
        labels = torch.cat([torch.ones(head.numel(), device=self.device),
                            torch.zeros(nh.numel(), device=self.device)])
        return torch.cat([head, nh]), torch.cat([tail, nt]), torch.cat([rel, nr]), labels

    # -- one epoch ----------------------------------------------------------------------
    def _make_batch(self, lo: int, size: int):
        """(heads, tails, rels, labels) of train columns ``order[lo : lo + size]`` plus their
        negatives.  Device sampler: one launch reading the position from ``self._cursor``;
        ``--torch_sampler``: the reference's op sequence on torch's RNG stream."""
        if self.device_sampler:
            from . import ops
            return ops.sample_batch(self.train_edge_index, self.train_edge_type, self._order, self._cursor, size,
                                    self.neg_sampler.num_neg_samples, self.neg_sampler.num_nodes, self._rng)
        idx = self._order.index_select(0, self._arange[:size] + self._cursor)
        return self._with_negatives(self.train_edge_index[0].index_select(0, idx),
                                    self.train_edge_index[1].index_select(0, idx),
                                    self.train_edge_type.index_select(0, idx))

    def _step(self, lo: int, size: int, accum: int = 1, update: bool = True):
        """One batch: assembly, forward over the train graph, BCE, backward and (when
        ``update``) clip + optimizer step; running loss / hits stay on the device.  ``lo`` is
        only documentation here - the position is whatever ``self._cursor`` holds."""
        heads, tails, rels, labels = self._make_batch(lo, size)
        fused = self._fused_bookkeeping
        if fused:
            # the criterion fused into the head; the launch that forms the mean also keeps the epoch's running sums
            # and moves the batch cursor on (device-side bookkeeping, no host sync, no elementwise launches)
            loss, scores = self.model.bce_loss(self.train_edge_index, self.train_edge_type, heads, tails, rels, labels,
                                               stats=(self._loss_sum, self._correct, self._cursor, size))
        else:
            scores = self.model(self.train_edge_index, self.train_edge_type, heads, tails, rels)
            loss = self.criterion(scores, labels)
        # d(loss / accum) / d loss as a kept device scalar: `.backward()` would launch a fill for its implicit ones
        # tensor (and a division for the accumulation) in every step
        seed = self._grad_seed.get(accum)
        if seed is None:
            seed = self._grad_seed[accum] = torch.full((), 1.0 / accum, device=self.device)
        loss.backward(seed)
        if update:
            self._clip_and_update()
        if not fused:
            with torch.no_grad():
                self._loss_sum += loss.detach().double() * labels.numel()
                self._correct += ((scores.detach() > 0) == (labels > 0.5)).sum()
        return heads, tails, rels, labels, loss.detach()

    def _clip_and_update(self) -> None:
        """``clip_grad_norm_`` + ``optimizer.step()`` (train.py:311-317).  On the GPU: two launches
        (``rgcn_adam_clip_step``) on the optimizer's own state tensors, so checkpoints and a later
        ``optimizer.step()`` see the same state; ``--torch_optimizer`` keeps torch's kernels."""
        if not self.fused_update:
            if self.args.grad_clip > 0:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.args.grad_clip)
            self.optimizer.step()
            return
        from . import ops
        group = self.optimizer.param_groups[0]
        params = [p for p in group["params"] if p.grad is not None]
        state = self.optimizer.state
        for p in params:
            if len(state[p]) == 0:                     # what torch's Adam would create lazily
                state[p]["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        beta1, beta2 = group["betas"]
        # The update also leaves max |p| of every matrix it writes (the embedding table, W, root) in an amax buffer of
        # the trainer's: the optimizer is their only writer, so the NEXT step's encoder takes its operand scales from
        # there (ops.amax_hint) and its first launch splits the weights without scanning 8 MB + the weights first.
        amax = [self._amax_of(p) for p in params]
        ops.adam_clip_step([p.data for p in params], [p.grad for p in params], [state[p]["exp_avg"] for p in params],
                           [state[p]["exp_avg_sq"] for p in params], [state[p]["step"] for p in params],
                           group["lr"], beta1, beta2, group["eps"], group["weight_decay"],
                           adamw=isinstance(self.optimizer, torch.optim.AdamW), max_norm=self.args.grad_clip,
                           amax_out=amax)
        for p, a in zip(params, amax):
            if a is not None:
                ops.set_amax_hint(p, a)              # valid until the tensor is modified through torch (p._version)

    def _amax_of(self, p):
        """the amax buffer the fused update publishes max |p| into (matrices only; zeroed once, kept for the run)"""
        if p.dim() < 2 or p.dtype != torch.float32:
            return None
        store = self.__dict__.setdefault("_param_amax", {})
        buf = store.get(id(p))
        if buf is None:
            from . import ops
            buf = store[id(p)] = ops.amax_buffer(p.device)[0]
        return buf

    def _capture_step(self, batch: int) -> None:
        """Record ``_step`` on the batch at ``self._cursor`` (``order``, ``cursor``, the RNG
        state and the running sums are device-resident and keep their addresses), so a
        full-size batch is one graph replay with no host work besides setting the cursor.
        Called after at least one eager step (graph bucketed, optimizer state allocated)."""
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            self._static = self._step(0, batch)
        self.optimizer.zero_grad(set_to_none=True)      # eager steps get fresh grads; replays use the graph's own
        self._graph, self._graph_batch = graph, batch

    def train_epoch(self, on_step: Optional[Callable] = None, max_steps: Optional[int] = None):
        """-> (mean loss per sample, accuracy).  ``on_step(heads, tails, rels, labels, loss)`` is
        an instrumentation hook (tests); ``max_steps`` truncates the epoch."""
        self.model.train()
        accum = max(1, getattr(self.args, "gradient_accumulation_steps", 1))
        bsz, e = self.args.batch_size, self.train_edge_index.size(1)
        if self._loss_sum is None:
            self._loss_sum = torch.zeros((), device=self.device, dtype=torch.float64)
            self._correct = torch.zeros((), device=self.device, dtype=torch.int64)
            self._order = torch.empty(e, device=self.device, dtype=torch.int64)
            self._cursor = torch.zeros(1, device=self.device, dtype=torch.int64)
            self._arange = torch.arange(bsz, device=self.device)
            # Philox key = the run's seed, stream = epoch (device sampler)
            self._rng = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64).to(self.device)
        self._loss_sum.zero_()
        self._correct.zero_()
        self._cursor.zero_()
        self._rng[1] += 1
        # the permutation is drawn on the host like the reference's torch.randperm(num_edges)
        self._order.copy_(torch.randperm(e))
        steps = -(-e // bsz)
        if max_steps is not None:
            steps = min(steps, max_steps)
        seen = 0
        self.optimizer.zero_grad(set_to_none=True)
        for step in range(steps):
            lo, hi = step * bsz, min((step + 1) * bsz, e)
            if not self._fused_bookkeeping:
                self._cursor.fill_(lo)               # (the fused bookkeeping launch moves the cursor itself)
            replayable = self.use_hip_graph and hi - lo == bsz
            if replayable and step >= 1 and (self._graph is None or self._graph_batch != bsz):
                self._capture_step(bsz)
            if replayable and self._graph is not None and self._graph_batch == bsz:
                self._graph.replay()
                out = self._static
            else:
                last = (step + 1) % accum == 0 or step + 1 == steps
                out = self._step(lo, hi - lo, accum=accum, update=last)
                if last:
                    self.optimizer.zero_grad(set_to_none=True)
            seen += (hi - lo) * (1 + self.neg_sampler.num_neg_samples)
            if on_step is not None:
                on_step(*out)
        result = (self._loss_sum / max(seen, 1)).item(), self._correct.item() / max(seen, 1)
        ops.check_indices(self.device)       # an id outside the embedding table anywhere in the epoch: IndexError, here
        return result

    @torch.no_grad()
    def validate(self):
        """Validation triples scored with messages passed over the FULL graph."""
        self.model.eval()
        loss_sum = torch.zeros((), device=self.device, dtype=torch.float64)
        correct = torch.zeros((), device=self.device, dtype=torch.int64)
        seen = 0
        # The reference re-runs the encoder for every validation batch (train.py:389-395); in eval mode nothing it
        # reads changes between the batches, so the node embeddings are computed ONCE here - the same rows - and every
        # batch is the head's two launches (scores + loss, then mean / running sums).  Negatives: the reference's
        # sampler on torch's RNG stream, as there.
        fused = self._fused_bookkeeping and hasattr(self.model, "encoder") and hasattr(self.model, "decoder")
        emb = self.model.encoder(self.full_edge_index, self.full_edge_type) if fused else None
        for head, tail, rel in self._batches(self.val_edge_index, self.val_edge_type, shuffle=False):
            heads, tails, rels, labels = self._with_negatives(head, tail, rel)
            if fused:
                self.model.decoder.bce_loss(emb, heads, tails, rels, labels, stats=(loss_sum, correct, None, 0))
            else:
                scores = self.model(self.full_edge_index, self.full_edge_type, heads, tails, rels)
                loss_sum += self.criterion(scores, labels).double() * labels.numel()
                correct += ((scores > 0) == (labels > 0.5)).sum()
            seen += labels.numel()
        result = (loss_sum / max(seen, 1)).item(), correct.item() / max(seen, 1)
        ops.check_indices(self.device)
        return result

    # -- checkpoints (same keys as train.py:431-442) --------------------------------------
    def save_checkpoint(self, epoch: int, is_best: bool = False, is_final: bool = False,
                        filename: Optional[str] = None) -> Path:
        state = {"epoch": epoch, "model_state_dict": self.model.state_dict(),
                 "optimizer_state_dict": self.optimizer.state_dict(),
                 "best_val_loss": self.best_val_loss, "best_val_acc": self.best_val_acc,
                 "train_losses": self.train_losses, "val_losses": self.val_losses,
                 "train_accs": self.train_accs, "val_accs": self.val_accs, "args": self.args}
        if is_best:
            path = self.model_dir / "best_model.pt"
        elif is_final:
            path = self.model_dir / "final_model.pt"
        else:
            path = self.checkpoint_dir / (filename or f"checkpoint_epoch_{epoch}.pt")
        torch.save(state, path)
        return path

    def train(self) -> None:
        start = time.time()
        epoch = 0
        for epoch in range(1, self.args.epochs + 1):
            t0 = time.time()
            tr_loss, tr_acc = self.train_epoch()
            va_loss, va_acc = self.validate()
            self.train_losses.append(tr_loss); self.train_accs.append(tr_acc)
            self.val_losses.append(va_loss); self.val_accs.append(va_acc)
            logger.info("Epoch %d/%d | Time: %.2fs | Train Loss: %.4f | Train Acc: %.4f | Val Loss: %.4f | "
                        "Val Acc: %.4f", epoch, self.args.epochs, time.time() - t0, tr_loss, tr_acc, va_loss, va_acc)
            is_best = va_loss < self.best_val_loss
            if is_best:
                self.best_val_loss = va_loss
            self.best_val_acc = max(self.best_val_acc, va_acc)
            if epoch % self.args.save_every == 0 or is_best:
                self.save_checkpoint(epoch, is_best=is_best)
            patience = self.args.early_stopping
            if patience > 0 and len(self.val_losses) > patience:
                recent = self.val_losses[-patience:]
                if all(v >= recent[0] for v in recent):
                    logger.info("Early stopping at epoch %d", epoch)
                    break
        logger.info("Training completed in %.2fs | best val loss %.4f | best val acc %.4f",
                    time.time() - start, self.best_val_loss, self.best_val_acc)
        self.save_checkpoint(epoch, is_final=True)


# ------------------------------------------------------------------------------------------
# data / CLI
# ------------------------------------------------------------------------------------------
def filter_edges(data: Dict, num_nodes: int, name: str = "") -> Dict:
    """Drop columns whose endpoints are >= num_nodes (the reference's host-side filter; the
    processed files hold a few such ids because of the mappings quirk)."""
    ei, et = data["edge_index"], data["edge_type"]
    keep = (ei[0] < num_nodes) & (ei[1] < num_nodes)
    dropped = int((~keep).sum())
    if dropped:
        logger.warning("%s: filtered %d invalid edges (%.2f%%)", name, dropped, 100.0 * dropped / ei.size(1))
        data = dict(data, edge_index=ei[:, keep], edge_type=et[keep])
    return data


def load_data(data_dir: str, bucket_cache: bool = False):
    """``{train,val,test}_data.pt``, ``full_graph.pt``: dicts {edge_index int64[2,E], edge_type
    int64[E], num_nodes, num_relations} (``preprocess.py:256-261``); tensors only, so they are
    read with ``weights_only=True``.  ``mappings.pt`` holds Python dicts and is not needed to train.
    ``bucket_cache``: keep the bucketed CSR-by-relation structure of the two message-passing
    graphs in ``<name>.bucketed.pt`` next to them (checked against the columns on load)."""
    root = Path(data_dir)
    files = (("train", "train_data.pt"), ("val", "val_data.pt"), ("test", "test_data.pt"), ("full", "full_graph.pt"))
    parts = {k: torch.load(root / f, weights_only=True) for k, f in files}
    n = parts["train"]["num_nodes"]
    parts = {k: filter_edges(v, n, k) for k, v in parts.items()}
    if bucket_cache:
        for k, f in files:
            parts[k] = dict(parts[k], sidecar=str(root / f.replace(".pt", ".bucketed.pt")))
    return parts["train"], parts["val"], parts["full"], parts["test"]


def synthetic_data(num_edges: int = 100_000, seed: int = 42, holdout: float = 0.15):
    """PrimeKG-shaped random graph split the way ``preprocess.py`` splits: drug-gene pairs
    (relation 0) are the prediction targets; ``holdout`` of them (the reference holds out
    7,696 of 51,306) leave the training graph for validation, both directions of a pair together."""
    from . import synth
    ei, et, n, r = synth.primekg_like(num_edges=num_edges, seed=seed)
    pair_is_target = et[0::2] == 0
    gen = torch.Generator().manual_seed(seed + 1)
    val_pairs = pair_is_target & (torch.rand(pair_is_target.numel(), generator=gen) < holdout)
    col_is_val = val_pairs.repeat_interleave(2)
    mk = lambda m: {"edge_index": ei[:, m].contiguous(), "edge_type": et[m].contiguous(),   # noqa: E731
                    "num_nodes": n, "num_relations": r}
    full = {"edge_index": ei, "edge_type": et, "num_nodes": n, "num_relations": r}
    return mk(~col_is_val), mk(col_is_val), full, mk(col_is_val)


def create_model(num_nodes: int, num_relations: int, args: argparse.Namespace) -> DrugDiseaseModel:
    model = DrugDiseaseModel(num_nodes=num_nodes, num_relations=num_relations,
                             embedding_dim=args.embedding_dim, hidden_dim=args.hidden_dim,
                             dropout=args.dropout, decoder_dropout=args.decoder_dropout,
                             num_bases=args.num_bases,
                             gather_dtype=torch.float16 if getattr(args, "fp16_gather", False) else None)
    logger.info("Model created with %s parameters",
                f"{sum(p.numel() for p in model.parameters() if p.requires_grad):,}")
    return model


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train the R-GCN link predictor on MI355X")
    p.add_argument("--data_dir", type=str, default="data/processed")
    p.add_argument("--output_dir", type=str, default="output")
    p.add_argument("--checkpoint_dir", type=str, default=None, help="[deprecated] use --output_dir")
    p.add_argument("--embedding_dim", type=int, default=64)
    p.add_argument("--hidden_dim", type=int, default=128)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--decoder_dropout", type=float, default=0.1)
    p.add_argument("--num_bases", type=int, default=None)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--optimizer", type=str, default="adam", choices=["adam", "adamw"])
    p.add_argument("--num_neg_samples", type=int, default=1)
    p.add_argument("--grad_clip", type=float, default=1.0)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--save_every", type=int, default=10)
    p.add_argument("--early_stopping", type=int, default=0)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--fp16_gather", action="store_true",
                   help="gather neighbour rows from an fp16 copy of the feature table (fp32 accumulate)")
    p.add_argument("--bucket_cache", action="store_true",
                   help="persist / reuse the bucketed graph structure next to the .pt files in --data_dir")
    p.add_argument("--torch_optimizer", action="store_true",
                   help="clip and update with torch's own kernels instead of the two-launch fused update")
    p.add_argument("--torch_sampler", action="store_true",
                   help="draw negatives with the reference's torch.rand/randint sequence instead of the "
                        "one-kernel device sampler")
    p.add_argument("--no_hip_graph", action="store_true",
                   help="launch every training step eagerly instead of replaying one captured HIP graph")
    p.add_argument("--synthetic", action="store_true",
                   help="train on a PrimeKG-shaped synthetic graph instead of --data_dir")
    p.add_argument("--synthetic_edges", type=int, default=100_000)
    return p


def parse_args(argv=None) -> argparse.Namespace:
    return build_parser().parse_args(argv)


def set_seed(seed: int) -> None:
    torch.manual_seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None) -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    args = parse_args(argv)
    if args.checkpoint_dir is not None:
        logger.warning("--checkpoint_dir is deprecated; using it as --output_dir")
        args.output_dir = args.checkpoint_dir
    set_seed(args.seed)
    device = torch.device(args.device)
    if args.synthetic:
        train_data, val_data, full_graph, _ = synthetic_data(args.synthetic_edges, args.seed)
    else:
        train_data, val_data, full_graph, _ = load_data(args.data_dir, args.bucket_cache)
    model = create_model(train_data["num_nodes"], train_data["num_relations"], args)
    Trainer(model, train_data, val_data, full_graph, device, args).train()


if __name__ == "__main__":
    main()

"""Synthetic graphs of the shapes BASELINE.json names (no dataset can be fetched, and the
reference's ``full_graph.pt`` / ``train_data.pt`` are missing from its mount).

``primekg_like``: the PrimeKG drug/gene/disease subgraph shape the reference trains on -
N = 30,926 with type-sorted ids (disease, drug, gene/protein: counts from the reference's
``data/processed/statistics.csv``), relation ids 0 = drug-gene, 1 = gene-disease,
2 = gene-gene in the proportions of that file (51,306 / 160,822 / 642,150), every
undirected pair emitted as two adjacent reverse columns the way
``src/preprocess.py:228-234`` does, endpoints type-constrained per relation and drawn
from a Zipf-like (1/rank) popularity so the degree distribution is heavy tailed and
duplicate columns occur, as in the real data.

``uniform_graph``: ``torch.randint`` endpoints and types (the reference's own self-test
recipe, ``src/models/rgcn.py:443-444``) for configs C1 and C4.
"""
from __future__ import annotations

from typing import Tuple

import torch

# node-type layout: disease [0, 5593), drug [5593, 11875), gene/protein [11875, 30926).
# statistics.csv counts 5,593 / 6,282 / 19,093 = 30,968 (id, name, type) triples, but the
# graph has num_nodes = 30,926 (val_data.pt / test_data.pt; preprocess.py:156-165 keys nodes
# by (id, type)), so the gene range is cut at 30,926 as SURVEY.md section 8d lays it out.
PRIMEKG_NODES = 30_926
N_DISEASE, N_DRUG = 5593, 6282
N_GENE = PRIMEKG_NODES - N_DISEASE - N_DRUG            # 19,051
PRIMEKG_EDGES = 849_456                                # README.md:47 / BASELINE.json
PRIMEKG_TRAIN_EDGES = 1_677_772                        # 2 x 838,886 train rows
REL_ROWS = (51_306, 160_822, 642_150)                  # drug-gene, gene-disease, gene-gene
_TYPE_RANGE = {"disease": (0, N_DISEASE), "drug": (N_DISEASE, N_DISEASE + N_DRUG),
               "gene": (N_DISEASE + N_DRUG, PRIMEKG_NODES)}
_REL_TYPES = (("drug", "gene"), ("gene", "disease"), ("gene", "gene"))


def _zipf_nodes(kind: str, count: int, gen: torch.Generator, exponent: float) -> torch.Tensor:
    lo, hi = _TYPE_RANGE[kind]
    n = hi - lo
    ranks = torch.arange(1, n + 1, dtype=torch.float64)
    prob = ranks.pow(-exponent)
    prob = prob[torch.randperm(n, generator=gen)]        # popularity shuffled within the type
    return torch.multinomial(prob, count, replacement=True, generator=gen) + lo


def primekg_like(num_edges: int = PRIMEKG_EDGES, seed: int = 42,
                 exponent: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
    """-> (edge_index int64[2, E], edge_type int64[E], num_nodes, num_relations) on CPU."""
    if num_edges % 2:
        raise ValueError("num_edges must be even (reverse pairs)")
    gen = torch.Generator().manual_seed(seed)
    pairs = num_edges // 2
    total = float(sum(REL_ROWS))
    per_rel = [int(round(pairs * c / total)) for c in REL_ROWS]
    per_rel[2] += pairs - sum(per_rel)
    us, vs, ts = [], [], []
    for rel, ((ka, kb), cnt) in enumerate(zip(_REL_TYPES, per_rel)):
        us.append(_zipf_nodes(ka, cnt, gen, exponent))
        vs.append(_zipf_nodes(kb, cnt, gen, exponent))
        ts.append(torch.full((cnt,), rel, dtype=torch.int64))
    u, v, t = torch.cat(us), torch.cat(vs), torch.cat(ts)
    order = torch.randperm(pairs, generator=gen)
    u, v, t = u[order], v[order], t[order]
    edge_index = torch.empty(2, num_edges, dtype=torch.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = u, v      # column 2k   : u -> v
    edge_index[0, 1::2], edge_index[1, 1::2] = v, u      # column 2k+1 : v -> u
    edge_type = t.repeat_interleave(2)
    return edge_index, edge_type, PRIMEKG_NODES, 3


def uniform_graph(num_nodes: int, num_edges: int, num_relations: int,
                  seed: int = 42) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
    gen = torch.Generator().manual_seed(seed)
    edge_index = torch.randint(0, num_nodes, (2, num_edges), generator=gen)
    edge_type = torch.randint(0, num_relations, (num_edges,), generator=gen)
    return edge_index, edge_type, num_nodes, num_relations

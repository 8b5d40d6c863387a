"""On-disk graph format either side of the path (SURVEY.md section 8f "next" row 3).

The reference's ``src/preprocess.py`` turns the filtered PrimeKG edge table into index maps
(``build_mappings``, ``preprocess.py:142-187``) and into the dict
``{edge_index int64[2, E], edge_type int64[E], num_nodes, num_relations}`` that ``train.py`` /
``evaluate.py`` load (``convert_to_pyg_format``, ``preprocess.py:189-263``), walking the table
with ``DataFrame.iterrows`` twice (minutes for the 854k-row table).  Here both are column
operations (seconds), producing the same maps and the same tensors - including the format's
quirks, which the rest of the pipeline depends on:

* a node is the triple (str(id), name, type); triples are sorted by (type, id, name) and
  enumerated, but ``node2idx`` is keyed by (id, type) only, so when two triples share a key the
  LAST index wins and ``len(node2idx) < len(idx2node)``;
* ``num_nodes = len(node2idx)``, and an edge whose endpoint index is ``>= num_nodes`` is dropped;
* every table row becomes two adjacent columns, (src, tgt) then (tgt, src), same relation id.

``save_graph`` / ``load_graph`` write and read the dict with tensors only (``weights_only=True``).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import torch

NODE_COLUMNS = (("x_id", "x_name", "x_type"), ("y_id", "y_name", "y_type"))


def build_mappings(df) -> Tuple[Dict, Dict, Dict, Dict]:
    """-> (node2idx {(id, type): idx}, idx2node {idx: (id, name, type)}, relation2idx, idx2relation)."""
    import pandas as pd
    parts = []
    for cid, cname, ctype in NODE_COLUMNS:
        parts.append(pd.DataFrame({"id": df[cid].astype(str).to_numpy(), "name": df[cname].to_numpy(),
                                   "type": df[ctype].to_numpy()}))
    nodes = pd.concat(parts, ignore_index=True).drop_duplicates()
    nodes = nodes.sort_values(["type", "id", "name"], kind="stable").reset_index(drop=True)
    ids, names, types = nodes["id"].tolist(), nodes["name"].tolist(), nodes["type"].tolist()
    idx2node = {i: (ids[i], names[i], types[i]) for i in range(len(ids))}
    node2idx = {(ids[i], types[i]): i for i in range(len(ids))}          # a repeated key keeps its last index
    relations = sorted(df["relation_standard"].unique())
    relation2idx = {r: i for i, r in enumerate(relations)}
    idx2relation = {i: r for i, r in enumerate(relations)}
    return node2idx, idx2node, relation2idx, idx2relation


def convert_to_pyg_format(df, node2idx: Dict, relation2idx: Dict) -> Dict:
    """-> {"edge_index", "edge_type", "num_nodes", "num_relations"} (reverse pairs adjacent)."""
    import pandas as pd
    num_nodes = len(node2idx)
    keys = pd.MultiIndex.from_tuples(list(node2idx.keys()), names=["id", "type"])
    table = pd.Series(np.fromiter(node2idx.values(), dtype=np.int64, count=num_nodes), index=keys)

    def lookup(cid, ctype) -> np.ndarray:
        want = pd.MultiIndex.from_arrays([df[cid].astype(str).to_numpy(), df[ctype].to_numpy()])
        return table.reindex(want).to_numpy(dtype=np.float64, na_value=np.nan)     # nan = node not in the mapping

    src, tgt = lookup("x_id", "x_type"), lookup("y_id", "y_type")
    rel = df["relation_standard"].map(relation2idx).to_numpy(dtype=np.float64, na_value=np.nan)
    if np.isnan(rel).any():
        raise KeyError("relation_standard holds a relation that is not in relation2idx")
    ok = ~np.isnan(src) & ~np.isnan(tgt)
    ok &= (np.nan_to_num(src, nan=-1) < num_nodes) & (np.nan_to_num(tgt, nan=-1) < num_nodes)
    ok &= (np.nan_to_num(src, nan=-1) >= 0) & (np.nan_to_num(tgt, nan=-1) >= 0)
    s, t, r = src[ok].astype(np.int64), tgt[ok].astype(np.int64), rel[ok].astype(np.int64)
    edge_index = np.empty((2, 2 * s.size), dtype=np.int64)
    edge_index[0, 0::2], edge_index[1, 0::2] = s, t          # column 2k   : src -> tgt
    edge_index[0, 1::2], edge_index[1, 1::2] = t, s          # column 2k+1 : tgt -> src
    return {"edge_index": torch.from_numpy(edge_index), "edge_type": torch.from_numpy(np.repeat(r, 2)),
            "num_nodes": num_nodes, "num_relations": len(relation2idx)}


def save_graph(path, data: Dict) -> None:
    """The dict ``train.py:563-567`` / ``evaluate.py:744-746`` load; tensors and ints only."""
    out = {"edge_index": data["edge_index"].to(torch.int64).contiguous(),
           "edge_type": data["edge_type"].to(torch.int64).contiguous(),
           "num_nodes": int(data["num_nodes"]), "num_relations": int(data["num_relations"])}
    if out["edge_index"].dim() != 2 or out["edge_index"].size(0) != 2 or out["edge_type"].shape != (out["edge_index"].size(1),):
        raise ValueError("edge_index must be [2, E] and edge_type [E]")
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(out, path)


def load_graph(path) -> Dict:
    data = torch.load(path, weights_only=True)
    missing = {"edge_index", "edge_type", "num_nodes", "num_relations"} - set(data)
    if missing:
        raise ValueError(f"{path}: not a graph dict (missing {sorted(missing)})")
    return data

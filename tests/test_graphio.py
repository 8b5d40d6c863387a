"""On-disk graph format (primekg_rgcn_linkprediction_amd/graphio.py), SURVEY section 8f next row 3:
the vectorised table -> graph conversion against the row-by-row restatement of preprocess.py."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import graphio_oracle as GO
from primekg_rgcn_linkprediction_amd import graphio as G


def _table(rows, seed):
    rng = np.random.default_rng(seed)
    types = np.array(["disease", "drug", "gene/protein"])
    rels = np.array(["drug_gene", "gene_disease", "gene_gene"])
    n_ids = 60

    def side():
        ids = rng.integers(0, n_ids, rows)
        t = types[rng.integers(0, 3, rows)]
        # the same (id, type) under two different names now and then: the key collision the format has
        names = np.where(rng.random(rows) < 0.15, "alias_", "name_") + pd.Series(ids).astype(str).to_numpy() + "_" + t
        return ids, names, t
    xi, xn, xt = side()
    yi, yn, yt = side()
    return pd.DataFrame({"x_id": xi, "x_name": xn, "x_type": xt, "y_id": yi, "y_name": yn, "y_type": yt,
                         "relation_standard": rels[rng.integers(0, 3, rows)]})


@pytest.mark.parametrize("rows,seed", [(1, 0), (50, 1), (3000, 2)])
def test_vectorised_conversion_equals_the_row_loop(rows, seed):
    df = _table(rows, seed)
    got_maps, want_maps = G.build_mappings(df), GO.build_mappings_ref(df)
    for g, w in zip(got_maps, want_maps):
        assert g == w
    node2idx, idx2node, rel2idx, _ = got_maps
    if rows >= 3000:
        assert len(node2idx) < len(idx2node)              # aliases collapsed: indices beyond num_nodes exist
    got, want = G.convert_to_pyg_format(df, node2idx, rel2idx), GO.convert_to_pyg_format_ref(df, node2idx, rel2idx)
    assert got["num_nodes"] == want["num_nodes"] and got["num_relations"] == want["num_relations"]
    assert torch.equal(got["edge_index"], want["edge_index"]) and torch.equal(got["edge_type"], want["edge_type"])
    assert got["edge_index"].dtype == torch.int64 and got["edge_index"].is_contiguous()
    ei = got["edge_index"]
    assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0)) and int(ei.max()) < got["num_nodes"]


def test_rows_with_unmapped_nodes_are_dropped_and_files_round_trip(tmp_path):
    df = _table(200, 5)
    node2idx, _, rel2idx, _ = G.build_mappings(df)
    extra = df.iloc[:3].copy()
    extra["x_id"] = 10_000                                  # a node the mapping has never seen
    both = pd.concat([df, extra], ignore_index=True)
    got, want = G.convert_to_pyg_format(both, node2idx, rel2idx), GO.convert_to_pyg_format_ref(both, node2idx, rel2idx)
    assert torch.equal(got["edge_index"], want["edge_index"]) and torch.equal(got["edge_type"], want["edge_type"])
    with pytest.raises(KeyError):
        G.convert_to_pyg_format(df.assign(relation_standard="unknown"), node2idx, rel2idx)
    G.save_graph(tmp_path / "d" / "train_data.pt", got)
    back = G.load_graph(tmp_path / "d" / "train_data.pt")
    assert torch.equal(back["edge_index"], got["edge_index"]) and back["num_nodes"] == got["num_nodes"]
    from primekg_rgcn_linkprediction_amd import train as T
    assert T.filter_edges(back, back["num_nodes"], "x")["edge_index"].size(1) == got["edge_index"].size(1)
    torch.save({"edge_index": got["edge_index"]}, tmp_path / "bad.pt")
    with pytest.raises(ValueError):
        G.load_graph(tmp_path / "bad.pt")
    with pytest.raises(ValueError):
        G.save_graph(tmp_path / "x.pt", dict(got, edge_type=got["edge_type"][:-1]))

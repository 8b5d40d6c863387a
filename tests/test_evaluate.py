"""Evaluation protocol (primekg_rgcn_linkprediction_amd/evaluate.py), SURVEY 8f next row 2."""
import numpy as np
import pytest
import torch

from conftest import load_golden, need_gpu
from primekg_rgcn_linkprediction_amd import DrugDiseaseModel
from primekg_rgcn_linkprediction_amd.evaluate import ModelEvaluator

pytestmark = pytest.mark.gpu


def _reference_ranks(scores, tails):
    """evaluate.py:266-274 verbatim in spirit: position of the true tail in the descending sort"""
    out = []
    for i, t in enumerate(tails.tolist()):
        order = torch.argsort(scores[i], descending=True)
        out.append((order == t).nonzero(as_tuple=True)[0].item() + 1)
    return torch.tensor(out)


def test_fused_ranks_equal_argsort_ranks_on_the_reference_run_model():
    dev = need_gpu()
    z = load_golden("ref_model_eval.npz")
    sd = {k[4:].replace("__", "."): v for k, v in z.items() if k.startswith("sd__")}
    m = DrugDiseaseModel(100, 3, 64, 128)
    m.load_state_dict(sd)
    data = {"edge_index": z["edge_index"], "edge_type": z["edge_type"], "num_nodes": 100, "num_relations": 3}
    test = {"edge_index": torch.stack([z["head"], z["tail"]]), "edge_type": z["rel"], "num_nodes": 100,
            "num_relations": 3}
    ev = ModelEvaluator(m, test, data, dev)
    ranks = ev.tail_ranks().cpu()
    want = _reference_ranks(z["all_scores"], z["tail"])       # all_scores come from the REFERENCE's predict_all_tails
    assert torch.equal(ranks, want)
    metrics = ev.compute_ranking_metrics((1, 10, 50))
    assert abs(metrics["mrr"] - float(np.mean(1.0 / want.numpy()))) < 1e-12
    assert metrics["hits@50"] == float((want <= 50).float().mean())


def test_ranks_and_auc_on_a_larger_graph():
    dev = need_gpu()
    from primekg_rgcn_linkprediction_amd import train as T
    torch.manual_seed(0)
    tr, va, full, te = T.synthetic_data(num_edges=30000, seed=2)
    m = DrugDiseaseModel(full["num_nodes"], 3, 64, 128)
    ev = ModelEvaluator(m, te, full, dev, batch_size=1024)
    ranks = ev.tail_ranks()
    emb = ev.embeddings()
    h, t, r = ev.test_edge_index[0], ev.test_edge_index[1], ev.test_edge_type
    scores = m.decoder.score_all_tails(emb[h[:200]], r[:200], emb)
    true = scores.gather(1, t[:200].view(-1, 1))
    # tolerance band around the true score (the GEMM and the row-wise dot round differently)
    lo = (scores > true + 1e-5).sum(1) + 1
    hi = (scores > true - 1e-5).sum(1)
    assert bool(((ranks[:200] >= lo) & (ranks[:200] <= hi.clamp(min=1))).all())
    assert ranks.min() >= 1 and ranks.max() <= full["num_nodes"]
    out = ev.evaluate()
    assert 0.0 <= out["classification"]["auc_roc"] <= 1.0 and out["test_edges"] == te["edge_index"].size(1)
    s, l = ev.compute_scores_and_labels(num_neg_samples=2)
    assert s.shape == l.shape == (3 * out["test_edges"],) and l.sum() == out["test_edges"]


# ---------------------------------------------------------------- embedding consumers
def test_cosine_consumers_match_the_numpy_restatement():
    """pair scores (DistMult kernel on unit rows), score matrix and top-k drugs vs the oracle's
    numpy restatement of compare_methods.py:368-397 / case_studies.py:236-284."""
    dev = need_gpu()
    from oracle import rgcn_oracle as O
    from primekg_rgcn_linkprediction_amd import consumers as C
    gen = torch.Generator().manual_seed(11)
    emb = torch.randn(500, 128, generator=gen)
    emb[40] = emb[17]                                     # two candidates tie exactly
    drugs = torch.arange(10, 210).tolist()
    diseases = torch.arange(300, 420).tolist()
    want = O.cosine_scores_ref(emb.numpy(), drugs, diseases)
    e = emb.to(dev)
    got = C.cosine_score_matrix(e, drugs, diseases).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6
    pi = torch.randint(0, len(drugs), (1000,), generator=gen)
    pj = torch.randint(0, len(diseases), (1000,), generator=gen)
    pair = C.cosine_pair_scores(e, torch.tensor(drugs)[pi], torch.tensor(diseases)[pj]).cpu().numpy()
    assert np.abs(pair - want[pi.numpy(), pj.numpy()]).max() <= 2e-6
    unit = C.normalize_rows(e)
    for disease, k, thr in ((300, 10, 0.0), (333, 25, 0.5), (419, 5, 0.99)):
        ref = O.top_drugs_ref(emb.numpy(), disease, drugs, k, thr)
        top = C.predict_top_drugs(unit, disease, drugs, k, thr, normalized=True)
        assert len(top) == len(ref)
        for (gi, gs), (ri, rs) in zip(top, ref):
            assert abs(gs - rs) <= 2e-6
            assert gi == ri or abs(gs - rs) <= 1e-7       # order may swap only between fp-equal scores
    with pytest.raises(ValueError):
        C.cosine_pair_scores(e, [1, 2], [3])
    with pytest.raises(RuntimeError):
        C.cosine_pair_scores(emb, [1], [2])               # CPU tensor: no CPU fallback



def test_evaluate_cli_round_trip(tmp_path):
    """train 1 short epoch -> final_model.pt -> `evaluate.main` on files in the reference's on-disk
    format -> results.json / metrics_summary.txt with the reference's keys (results_final/results.json)."""
    need_gpu()
    import json
    from primekg_rgcn_linkprediction_amd import evaluate as E, train as T
    tr, va, full, te = T.synthetic_data(num_edges=20000, seed=4)
    data_dir = tmp_path / "processed"
    data_dir.mkdir()
    for name, d in (("train_data.pt", tr), ("val_data.pt", va), ("test_data.pt", te), ("full_graph.pt", full)):
        torch.save(d, data_dir / name)
    T.main(["--data_dir", str(data_dir), "--output_dir", str(tmp_path / "out"), "--epochs", "1", "--lr", "0.01",
            "--bucket_cache"])
    assert (data_dir / "train_data.bucketed.pt").exists() and (data_dir / "full_graph.bucketed.pt").exists()
    metrics = E.main(["--model_path", str(tmp_path / "out" / "models" / "final_model.pt"), "--data_dir", str(data_dir),
                      "--output_dir", str(tmp_path / "results"), "--k_values", "10", "50", "100"])
    saved = json.loads((tmp_path / "results" / "results.json").read_text())
    assert saved["metrics"] == metrics
    assert set(saved["metrics"]) == {"classification", "ranking", "test_edges", "num_nodes"}
    assert set(saved["metrics"]["classification"]) == {"auc_roc", "auc_pr", "precision", "recall", "f1_score", "threshold"}
    assert set(saved["metrics"]["ranking"]) == {"mrr", "mean_rank", "median_rank", "hits@10", "hits@50", "hits@100"}
    assert set(saved["model_info"]) >= {"checkpoint_path", "epoch", "num_nodes", "num_relations", "embedding_dim",
                                        "hidden_dim", "num_parameters", "best_val_loss", "best_val_acc"}
    assert saved["model_info"]["num_parameters"] == 2078208 and saved["metrics"]["test_edges"] == te["edge_index"].size(1)
    text = (tmp_path / "results" / "metrics_summary.txt").read_text()
    assert "EVALUATION RESULTS SUMMARY" in text and "Ranking Metrics:" in text and "hits@100" in text


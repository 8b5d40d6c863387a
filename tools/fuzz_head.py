"""Fuzz of the DistMult head (scores, BCE loss, deterministic backward) against float64 autograd: random batch sizes
around the kernel's block sizes, widths 4 ... 256, tables of 1 ... 5000 rows, heavy duplicates / one hub / all the same
row, head and tail from one table or two, operands with and without an index vector.

    python tools/fuzz_head.py [cases] [seed]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import LinkPredictor, distmult, ops  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 99)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def ids(b, rows, kind):
    i = torch.randint(0, rows, (b,), generator=gen)
    if kind == 1:
        i[torch.rand(b, generator=gen) < 0.5] = rnd(0, rows - 1)      # a hub
    elif kind == 2:
        i[:] = rnd(0, rows - 1)                                        # one row takes everything
    elif kind == 3:
        i = i % max(1, min(rows, 7))                                   # a handful of rows
    return i


t0, worst = time.time(), 0.0
for case in range(cases):
    b = [1, rnd(2, 70), rnd(250, 260), rnd(1000, 1030), rnd(2040, 2060), rnd(5000, 9000)][rnd(0, 5)]
    d = [4, 32, 36, 64, 128, 160, 256][rnd(0, 6)]
    rows, rows_t, rels = [1, rnd(2, 50), rnd(300, 5000)][rnd(0, 2)], rnd(1, 400), rnd(1, 6)
    shared, kind = bool(rnd(0, 1)), rnd(0, 3)
    label = f"case {case}: B={b} d={d} rows={rows} rows_t={rows_t} R={rels} shared={shared} kind={kind}"
    try:
        emb = torch.randn(rows, d, generator=gen)
        emb_t = emb if shared else torch.randn(rows_t, d, generator=gen)
        rel = torch.randn(rels, d, generator=gen)
        hi, ti, ri = ids(b, rows, kind), ids(b, emb_t.size(0), (kind + 1) % 4), ids(b, rels, rnd(0, 3))
        cot = torch.randn(b, generator=gen)
        labels = (torch.rand(b, generator=gen) > 0.5).float()
        e64, t64, r64 = (x.double().requires_grad_(True) for x in (emb, emb_t, rel))
        tt = e64 if shared else t64
        s64 = (e64[hi] * r64[ri] * tt[ti]).sum(1)
        loss64 = torch.nn.functional.binary_cross_entropy_with_logits(s64, labels.double())
        ((s64 * cot.double()).sum() + 3.0 * loss64).backward()
        e = emb.to(dev).requires_grad_(True)
        t = e if shared else emb_t.to(dev).requires_grad_(True)
        r = rel.to(dev).requires_grad_(True)
        sc = distmult(e, hi.to(dev), t, ti.to(dev), r, ri.to(dev))
        dec = LinkPredictor(rels, d, dropout=0.0).to(dev)
        with torch.no_grad():
            dec.relation_embeddings.weight.copy_(rel)
        if d % 4 == 0 and shared:
            loss, sc2 = dec.bce_loss(e, hi.to(dev), ti.to(dev), ri.to(dev), labels.to(dev))
            assert torch.equal(sc2, sc.detach())
            assert abs(float(loss.detach()) - float(loss64.detach())) <= 2e-6 * max(1.0, abs(float(loss64.detach()))), "loss"
            ((sc * cot.to(dev)).sum() + 3.0 * loss).backward()
            r_grad = r.grad + dec.relation_embeddings.weight.grad
        else:
            (sc * cot.to(dev)).sum().backward()
            r_grad = r.grad
            # redo the float64 side without the loss term
            for x in (e64, t64, r64):
                x.grad = None
            s64b = (e64[hi] * r64[ri] * tt[ti]).sum(1)
            (s64b * cot.double()).sum().backward()
        tol = lambda ref: 2e-5 * max(1e-30, float(ref.abs().max()))            # noqa: E731
        err = float((sc.detach().double().cpu() - s64.detach()).abs().max())
        assert err <= 1e-5 * max(1.0, float(s64.detach().abs().max())), f"scores {err:.2e}"
        pairs = [(e.grad, e64.grad), (r_grad, r64.grad)] + ([] if shared else [(t.grad, t64.grad)])
        for got, ref in pairs:
            ge = float((got.double().cpu() - ref).abs().max())
            worst = max(worst, ge / max(1e-30, float(ref.abs().max())))
            assert ge <= tol(ref), f"gradient {ge:.2e} of {float(ref.abs().max()):.2e}"
        print(f"ok   {label}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"FAIL {label}: {type(exc).__name__}: {exc}", flush=True)
ops.check_indices(dev)
print(f"{cases} cases in {time.time() - t0:.0f} s; worst gradient error {worst:.2e} of the gradient's largest entry")

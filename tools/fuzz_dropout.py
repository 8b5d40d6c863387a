"""Fuzz of the training-mode encoder (dropout between the layers): ``rgcn_encoder2(..., dropout_p)`` - conv1 + ReLU,
torch's own dropout kernel, conv2, with the dropout's backward folded into conv2's input-gradient epilogue and, from the
fourth step on, both forward halves and the backward issued natively - against the three separate ops
``conv2(F.dropout(relu(conv1(x)), p))`` under the same seed: the forward bit for bit, every gradient to 2e-6.  Random
graphs, widths, p in {0.1, 0.5, 0.8}, six steps each with fresh inputs.

    python tools/fuzz_dropout.py [cases] [seed]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, rgcn_encoder2  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 2468)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


t0, worst = time.time(), 0.0
for case in range(cases):
    n = [rnd(2, 70), rnd(120, 300), rnd(1000, 20000)][rnd(0, 2)]
    r = [1, 3, 5, 16][rnd(0, 3)]
    e = rnd(1, 200) if n < 100 else rnd(500, 100000)
    dims = [(64, 128, 128), (32, 32, 32), (128, 128, 64), (64, 256, 256), (64, 64, 32)][rnd(0, 4)]
    p = [0.1, 0.5, 0.8][rnd(0, 2)]
    ei = torch.randint(0, n, (2, e), generator=gen)
    if rnd(0, 1):
        ei[1, torch.rand(e, generator=gen) < 0.3] = rnd(0, n - 1)        # a hub
    et = torch.randint(0, r, (e,), generator=gen)
    label = f"case {case}: n={n} e={e} r={r} dims={dims} p={p}"
    try:
        torch.manual_seed(case)
        convs = [RGCNConv(dims[0], dims[1], r).to(dev), RGCNConv(dims[1], dims[2], r).to(dev)]
        for c in convs:
            c.bias.data.uniform_(-0.1, 0.1)
        eid, etd = ei.to(dev), et.to(dev)
        params = [q for c in convs for q in c.parameters()]
        for step in range(6):
            x = (torch.randn(n, dims[0], generator=gen) * (1 + step)).to(dev).requires_grad_(True)
            cot = torch.randn(n, dims[2], generator=gen).to(dev)
            torch.manual_seed(100 * case + step)
            got = rgcn_encoder2(x, eid, etd, convs[0], convs[1], dropout_p=p)
            g_got = torch.autograd.grad(got, [x] + params, cot)
            torch.manual_seed(100 * case + step)
            h = torch.relu(convs[0](x, eid, etd))
            want = convs[1](torch.nn.functional.dropout(h, p, True), eid, etd)
            g_want = torch.autograd.grad(want, [x] + params, cot)
            assert torch.equal(got, want), f"forward bits at step {step}"
            for a, b in zip(g_got, g_want):
                scale = float(b.abs().max())
                err = float((a - b).abs().max()) / max(scale, 1e-30)
                worst = max(worst, err)
                assert err <= 2e-6 or scale == 0.0, f"gradient {err:.2e} at step {step}"
        print(f"ok   {label}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"FAIL {label}: {type(exc).__name__}: {exc}", flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s; worst gradient distance between the fused node and the three ops {worst:.2e}")

"""Does any kernel read memory nobody wrote?  Run the two-layer encoder forward + backward twice - once on a fresh
allocator, once after filling the caching allocator's free blocks with a poison pattern - and compare every result
bit for bit.  python tools/poison_probe.py [half|split|fp32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, ops, rgcn_encoder2, synth  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "half"
gdt = torch.float16 if mode == "half" else None
if mode == "fp32":
    ops.GEMM_PRECISION = "fp32"
ei, et, n, r = synth.primekg_like(seed=42)
eid, etd = ei.to(dev), et.to(dev)
torch.manual_seed(5)
emb = torch.nn.init.xavier_uniform_(torch.empty(n, 64)).to(dev)
convs = [RGCNConv(64, 128, r, gather_dtype=gdt).to(dev), RGCNConv(128, 128, r, gather_dtype=gdt).to(dev)]
cot = (torch.randn(n, 128) * 1e-6).to(dev)


def poison(value):
    blocks = [torch.full((sz,), value, device=dev) for sz in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20,
                                                              1 << 22, 1 << 24, 1 << 25, 1 << 26) for _ in range(6)]
    del blocks


def step():
    e = emb.clone().requires_grad_(True)
    for c in convs:
        c.zero_grad()
    out = rgcn_encoder2(e, eid, etd, convs[0], convs[1])
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    res = {"out": out.detach().clone(), "emb.grad": e.grad.clone()}
    for i, c in enumerate(convs):
        for k, v in c.named_parameters():
            res[f"conv{i + 1}.{k}.grad"] = v.grad.clone()
    return res


want = step()
if os.environ.get("PROBE_EVAL_BETWEEN"):
    convs32 = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
    with torch.no_grad():
        rgcn_encoder2(emb, eid, etd, convs32[0], convs32[1])      # the no-grad (fused) encoder on the same graph
    got = step()
    bad = [k for k in want if not torch.equal(got[k], want[k])]
    print(f"{mode}: after a no-grad forward: " + ("all results bit-identical" if not bad else "DIFFERENT: " + ", ".join(
        f"{k} (max rel {float((got[k] - want[k]).abs().max() / want[k].abs().max()):.3e})" for k in bad)), flush=True)
for value in [float(v) for v in os.environ.get("PROBE_VALUES", "nan 3e38 1e-30 1.0").split()]:
    poison(value)
    got = step()
    bad = [k for k in want if not torch.equal(got[k], want[k])]
    print(f"{mode}: poison {value}: " + ("all results bit-identical" if not bad else "DIFFERENT: " + ", ".join(
        f"{k} (max rel {float((got[k] - want[k]).abs().max() / want[k].abs().max()):.3e})" for k in bad)), flush=True)

"""Diagnostic: host time to issue one eager encoder step (tiny graph, so the GPU is never the limit)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primekg_rgcn_linkprediction_amd import RGCNConv, rgcn_encoder2, synth

dev = torch.device("cuda:0")
ei, et, n, r = synth.uniform_graph(2000, 20000, 3, seed=1)
eid, etd = ei.to(dev), et.to(dev)
emb = torch.randn(n, 64, device=dev, requires_grad=True)
convs = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
cot = torch.randn(n, 128, device=dev)
params = [emb] + [p for c in convs for p in c.parameters()]

def step():
    out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
    for p in params:
        p.grad = None
    out.backward(cot)

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    step()
t_issue = (time.perf_counter() - t0) / 500
torch.cuda.synchronize()
print(f"host time per eager encoder step: {t_issue * 1e6:.0f} us")
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)

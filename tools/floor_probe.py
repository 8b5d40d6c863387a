"""Round 4 probe: the FIXED cost of every launch of the encoder step - the step on a 1k-node / 10k-edge graph, where no
kernel has real work.  Run under rocprofv3 --kernel-trace --stats: the per-kernel averages are the launches' floors
(their sum, plus ~1.5 us per boundary, is the 108 us the step costs however small the graph).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_floor -o p -- python3 tools/floor_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, rgcn_encoder2_step, synth  # noqa: E402

dev = torch.device("cuda:0")
ei, et, n, r = synth.uniform_graph(1000, 10000, 3, seed=1)
eid, etd = ei.to(dev), et.to(dev)
torch.manual_seed(0)
emb = torch.nn.init.xavier_uniform_(torch.empty(n, 64)).to(dev).requires_grad_(True)
convs = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
cot = torch.randn(n, 128, device=dev)
for _ in range(300):
    for p in [emb] + [q for c in convs for q in c.parameters()]:
        p.grad = None
    rgcn_encoder2_step(emb, eid, etd, convs[0], convs[1], cot)
torch.cuda.synchronize()

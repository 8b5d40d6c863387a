"""No-grad encoder on C4's graph (500k nodes / 20M edges / 16 relations, 64 -> 128 -> 128) on ONE GPU: the
destination-row-blocked path (conv.encoder2_eval) against one whole-graph block.  Prints time per forward and peak
memory; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` for the HBM-side bytes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, conv as C, ops, rgcn_encoder2, synth  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
ei, et, n, r = synth.uniform_graph(500_000, 20_000_000, 16, seed=42)
eid, etd = ei.to(dev), et.to(dev)
torch.manual_seed(0)
emb = torch.nn.init.xavier_uniform_(torch.empty(n, 64)).to(dev)
convs = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
ops.bucket(eid, etd, n, r)
for name, nbytes in (("blocked (32 MB aggregate blocks)", 32 << 20), ("one block (whole aggregate)", 1 << 42)):
    if mode not in ("both", name.split()[0]):
        continue
    C._EVAL_BLOCK_BYTES = nbytes
    with torch.no_grad():
        out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])         # builds the block structures once
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(5):
            out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
        torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 2-layer forward, peak "
          f"{torch.cuda.max_memory_allocated() / 2**30:.2f} GiB allocated, checksum {float(out.double().sum()):.6f}")

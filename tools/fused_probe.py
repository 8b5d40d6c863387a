"""No-grad encoder: the one-kernel layer (ops.layer_fwd_fused) against gather -> transform.  Time per 2-layer
forward, peak memory and per-launch event times of the fused kernel, at C2 (PrimeKG-shaped) or C4's graph on one GPU
(500k nodes / 20M edges / 16 relations).  Run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` for HBM bytes.
  python tools/fused_probe.py c2|c4 [fused|plain|both] [inline_limit ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, conv as C, ops, rgcn_encoder2, synth  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
limits = [int(a) for a in sys.argv[3:]] or [16]
if which == "c4":
    ei, et, n, r = synth.uniform_graph(500_000, 20_000_000, 16, seed=42)
else:
    ei, et, n, r = synth.primekg_like(seed=42)
eid, etd = ei.to(dev), et.to(dev)
torch.manual_seed(0)
emb = torch.nn.init.xavier_uniform_(torch.empty(n, 64)).to(dev)
convs = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
graph = ops.bucket(eid, etd, n, r)
reps = 5 if which == "c4" else 50


def run(tag):
    with torch.no_grad():
        out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        line = (f"{which} {tag}: {ms:.3f} ms per 2-layer forward, peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, "
                f"checksum {float(out.double().sum()):.6f}")
        if C._EVAL_FUSED:
            ops.FUSED_EVENTS = []
            rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
            torch.cuda.synchronize()
            line += "; fused launches " + ", ".join(f"{d_in}->{d_out}: {b.elapsed_time(e) * 1e3:.1f} us"
                                                    for _, _, _, _, _, d_in, d_out, b, e in ops.FUSED_EVENTS)
            ops.FUSED_EVENTS = None
    print(line, flush=True)
    return out


ref = None
if mode in ("both", "plain"):
    C._EVAL_FUSED = False
    ref = run("gather -> transform")
if mode in ("both", "fused"):
    C._EVAL_FUSED = True
    for limit in limits:
        C._EVAL_INLINE_LIMIT = limit
        plan = graph.fused_plan(limit)
        out = run(f"fused, inline limit {limit} ({plan.hub_rows} pre-aggregated segments, {plan.hub_edges} of {ei.size(1)} edges)")
        if ref is not None:
            print("   bit-identical to the two-launch path:", bool(torch.equal(out, ref)), flush=True)

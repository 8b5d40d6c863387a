"""Round 4 probe: how fast would the gathers be if every read hit the XCD's L2?  C2's graph with its SOURCE ids folded into
the first K rows of the table (same destinations, same segment lengths, same bytes gathered), K from the whole table down to
2,048 rows.  Timing only (HIP-graph replays)."""
import sys, torch
sys.path.insert(0, ".")
from primekg_rgcn_linkprediction_amd import ops, synth

dev = torch.device("cuda:0")
ei, et, n, r = synth.primekg_like()
ei, et = ei.to(dev), et.to(dev)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


print("rows read   table MB(d=128)   gather d=64   gather d=128   (us; destinations and segment lengths as C2)")
for k in (n, 16384, 8192, 4096, 2048):
    src = ei[0] % k
    graph = ops.bucket(torch.stack([src, ei[1]]), et, n, r)
    x64, x128 = torch.randn(n, 64, device=dev), torch.randn(n, 128, device=dev)
    t64 = timed(lambda: ops.aggregate(graph, x64))
    t128 = timed(lambda: ops.aggregate(graph, x128))
    print(f"{k:9d}   {k * 512 / 1e6:13.2f}   {t64:11.2f}   {t128:12.2f}")
    graph.destroy()

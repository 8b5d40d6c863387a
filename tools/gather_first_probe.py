"""Round 4 probe: conv2's forward gather over a transform-first table T2 [N, (R+1) * 128] (one weighted gather over all
relations + the root row, writing out [N, 128]) against the ordinary gather of h [N, 128] into agg2 [N, R * 128].
Timing only (HIP-graph replays), C2's graph."""
import sys, torch
sys.path.insert(0, ".")
from primekg_rgcn_linkprediction_amd import ops, synth

dev = torch.device("cuda:0")
ei, et, n, r = synth.primekg_like()
ei, et = ei.to(dev), et.to(dev)
graph = ops.bucket(ei, et, n, r)
d = 128
h = torch.randn(n, d, device=dev)
t2 = torch.randn(n, (r + 1) * d, device=dev)
src, dst = ei[0], ei[1]
cnt = torch.zeros(n * r, device=dev).index_add_(0, dst * r + et, torch.ones_like(src, dtype=torch.float32))
w = 1.0 / cnt[dst * r + et]
nodes = torch.arange(n, device=dev)
key = torch.cat([dst, nodes])
other = torch.cat([src * (r + 1) + et, nodes * (r + 1) + r])
weight = torch.cat([w, torch.ones(n, device=dev)])
merged = ops.BucketedGraph.from_shard(key, other, torch.zeros_like(key), n, n * (r + 1), 1, weight)
want = ops.aggregate(graph, h)
got = ops.aggregate(merged, t2.view(-1, d))
ref = torch.zeros(n, d, device=dev).index_add_(0, key, t2.view(-1, d)[other] * weight[:, None])
print("merged gather vs index_add:", float((got - ref).abs().max()))


def timed(fn, name, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"  {name:70s} {a.elapsed_time(b) / reps * 1e3:7.2f} us")


timed(lambda: ops.aggregate(graph, h), "gather of h [N,128] -> agg2 [N, 384]  (today's conv2 forward gather)")
timed(lambda: ops.aggregate(merged, t2.view(-1, d)), "gather of T2 [N*4, 128] over the merged structure -> out [N, 128]")
x64 = torch.randn(n, 64, device=dev)
timed(lambda: ops.aggregate(graph, x64), "gather of x [N,64] -> agg1 [N, 192]  (conv1 forward gather, for scale)")

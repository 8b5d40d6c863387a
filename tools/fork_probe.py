"""Round 4 probe: the pass's first launch (max |x| + the split weights: a latency chain on ~380 workgroups) BESIDE conv1's
gather (which needs neither) as two branches of a HIP graph, against the two launches back to back; then the whole forward
pass both ways.  Timing only (graph replays)."""
import sys, torch
sys.path.insert(0, ".")
from primekg_rgcn_linkprediction_amd import ops, synth, RGCNConv

dev = torch.device("cuda:0")
ei, et, n, r = synth.primekg_like()
ei, et = ei.to(dev), et.to(dev)
graph = ops.bucket(ei, et, n, r)
torch.manual_seed(0)
c1, c2 = RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)
x = torch.randn(n, 64, device=dev) * 0.1
layers = [(c1.weight.detach(), c1.root.detach()), (c2.weight.detach(), c2.root.detach())]
side = torch.cuda.Stream()


def first_launch():
    buf = torch.empty(2, ops.AMAX_FLOATS, device=dev)
    return buf, ops.absmax_and_split(x, buf[0], buf[1:2], layers)


def serial(rest):
    buf, packs = first_launch()
    agg, hubs = ops.aggregate_deferred(graph, x)
    return rest(buf, packs, agg, hubs)


def forked(rest):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        buf, packs = first_launch()
    agg, hubs = ops.aggregate_deferred(graph, x)
    main.wait_stream(side)
    return rest(buf, packs, agg, hubs)


def nothing(buf, packs, agg, hubs):
    return agg


def layer1(buf, packs, agg, hubs):
    return ops.transform_fwd(agg, x, layers[0][0], layers[0][1], c1.bias.detach(), relu=True, graph=graph, half=False,
                             amax=(buf[0], buf[0]), amax_out=buf[1], packed=packs[0], hubs=hubs)


def timed(fn, name, reps=40):
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
    print(f"  {name:78s} {best:7.2f} us")


o1 = serial(layer1)
o2 = forked(layer1)
torch.cuda.synchronize()
print("same bits:", bool(torch.equal(o1, o2)))
timed(lambda: first_launch(), "first launch alone")
timed(lambda: ops.aggregate_deferred(graph, x), "conv1's gather alone")
timed(lambda: serial(nothing), "first launch -> gather, one stream")
timed(lambda: forked(nothing), "first launch || gather (two graph branches, joined)")
timed(lambda: serial(layer1), "first launch -> gather -> conv1's transform, one stream")
timed(lambda: forked(layer1), "(first launch || gather) -> conv1's transform")

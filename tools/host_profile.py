"""Host time of one eager encoder step (forward + backward of ``rgcn_encoder2``), with and without the native step
(``ops.Region`` -> ``rgcn_sequence_run``).  Two measurements each:

  * C2 (30,926 / 849,456 / 3, 64 -> 128 -> 128): wall time per step of a long eager loop - the larger of what the host
    needs to issue a step and what the GPU needs to run it (0.28 ms of kernels);
  * the same encoder over a 1k-node / 10k-edge graph, whose kernels take a few microseconds each: the loop is then
    bound by the host alone, so its time per step IS the host cost of a step (launch list and argument handling do
    not depend on the graph's size).

    python tools/host_profile.py [n_lines_of_cProfile]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, ops, rgcn_encoder2, rgcn_encoder2_step, synth  # noqa: E402

dev = torch.device("cuda:0")


def make(ei, et, n, r):
    eid, etd = ei.to(dev), et.to(dev)
    torch.manual_seed(0)
    emb = torch.nn.init.xavier_uniform_(torch.empty(n, 64)).to(dev).requires_grad_(True)
    convs = [RGCNConv(64, 128, r).to(dev), RGCNConv(128, 128, r).to(dev)]
    cot = torch.randn(n, 128, device=dev)

    def step():
        out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
        out.backward(cot)

    def explicit():                                  # the same two passes without the autograd engine (round 4)
        for p in params:
            p.grad = None
        rgcn_encoder2_step(emb, eid, etd, convs[0], convs[1], cot)
    params = [emb] + [p for c in convs for p in c.parameters()]
    step.explicit = explicit
    return step


def per_step(step, steps):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    issue = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    return issue, (time.perf_counter() - t0) / steps


big = make(*synth.primekg_like(seed=42))
small = make(*synth.uniform_graph(1000, 10000, 3, seed=1))
for native in (True, False):
    ops.REGIONS = native
    label = "native step (one C call per pass)" if native else "wrappers (one ctypes call per launch)"
    issue, total = per_step(big, 200)
    print(f"{label}: C2 {issue * 1e6:.0f} us per step issued, {total * 1e6:.0f} us with the final sync")
    issue, total = per_step(small, 2000)
    print(f"{label}: 1k-node graph (host-bound) {total * 1e6:.0f} us per step = host cost of an eager encoder step")
ops.REGIONS = True
issue, total = per_step(big.explicit, 200)
print(f"explicit step (rgcn_encoder2_step: both passes, no autograd engine): C2 {issue * 1e6:.0f} us per step issued, {total * 1e6:.0f} us with the final sync")
issue, total = per_step(small.explicit, 2000)
print(f"explicit step (rgcn_encoder2_step: both passes, no autograd engine): 1k-node graph (host-bound) {total * 1e6:.0f} us per step")
# Is the small graph's loop really bound by the host?  Its DEVICE time per step = the same step as a replayed HIP graph
# (no host work between launches): twelve dependent launches have a floor of their own (each a launch boundary plus a
# prologue's latencies), and where that floor is close to the eager figure, the eager figure is not host cost - the
# "issued" time on C2 (the host running ahead of a busy GPU) is.
def replayed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps // 10 * 10)


dev_small, dev_big = replayed(small.explicit), replayed(big.explicit)
print(f"explicit step as a replayed HIP graph (device time alone): 1k-node graph {dev_small * 1e6:.0f} us per step, C2 {dev_big * 1e6:.0f} us")
for name, fn in (("autograd step", small), ("explicit step", small.explicit)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    print(f"---- cProfile of 2000 x {name} on the 1k-node graph")
    pstats.Stats(pr).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 25)

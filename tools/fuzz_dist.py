"""Fuzz of the node-partitioned form (dist.py) on ONE GPU: every rank's shard structures on the real kernels, the halo
exchange emulated by reading the rows it would deliver (``HaloPlan.emulate``), against the single-GPU path - bit for bit
in fp32 arithmetic, to 2e-6 in the default split arithmetic (each rank scales by the maximum of the rows it holds) -
over random graphs (empty relations, hubs, self loops, isolated halves, disjoint relation ranges), 2 ... 8 ranks,
node counts down to fewer nodes than ranks, with and without the interior / boundary split.

    python tools/fuzz_dist.py [cases] [seed]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import RGCNConv, ops  # noqa: E402
from primekg_rgcn_linkprediction_amd import dist as rdist  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


def graph(n, e, r):
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    et = torch.randint(0, r, (e,), generator=gen)
    kind = rnd(0, 4)
    if kind == 0 and r > 1:
        et[et == rnd(0, r - 1)] = 0
    elif kind == 1:
        dst[torch.rand(e, generator=gen) < 0.33] = rnd(0, n - 1)
        src[torch.rand(e, generator=gen) < 0.2] = rnd(0, n - 1)
    elif kind == 2:
        k = e // 4
        src[:k] = dst[:k]
    elif kind == 3:
        src, dst = src % max(1, n // 2), dst % max(1, n // 2)
    return torch.stack([src, dst]), et, kind


t0, worst = time.time(), 0.0
for case in range(cases):
    world = [2, 3, 4, 5, 8][rnd(0, 4)]
    n = [rnd(1, 12), rnd(30, 200), rnd(1000, 4000), rnd(8000, 20000)][rnd(0, 3)]
    r = [1, 3, 5, 16][rnd(0, 3)]
    e = rnd(1, 60) if n < 30 else rnd(200, 80000)
    d_in, d_out = [(64, 128), (128, 128), (32, 64), (128, 64)][rnd(0, 3)]
    precision = ["fp32", "split"][rnd(0, 1)]
    split = [None, True, False][rnd(0, 2)]
    ei, et, kind = graph(n, e, r)
    label = f"case {case}: P={world} n={n} e={e} r={r} {d_in}->{d_out} kind={kind} {precision} split={split}"
    ops.GEMM_PRECISION = precision
    try:
        x = torch.randn(n, d_in, generator=gen)
        g = torch.randn(n, d_out, generator=gen)
        torch.manual_seed(case)
        conv = RGCNConv(d_in, d_out, r).to(dev)
        conv.bias.data.uniform_(-0.1, 0.1)
        w, root, bias = conv.weight.detach(), conv.root.detach(), conv.bias.detach()
        graph1 = ops.BucketedGraph(ei.to(dev), et.to(dev), n, r)
        agg1 = ops.aggregate(graph1, x.to(dev))
        out1 = ops.transform_fwd(agg1, x.to(dev), w, root, bias)
        gx1 = ops.transform_bwd_input(ops.aggregate(graph1, g.to(dev), transposed=True), g.to(dev), w, root)
        gw1, groot1, gbias1 = ops.transform_bwd_params(agg1, x.to(dev), g.to(dev), r)
        backend = rdist.HipBackend()
        part = rdist.NodePartition(ei, n, world)
        xd, gd = x.to(dev), g.to(dev)
        outs, gxs, gw, groot, gbias = [], [], 0, 0, 0
        for k in range(world):
            shard = rdist.RankShard(part, ei, et, r, k, dev, backend, split=split)
            x_own, g_own = part.shard_rows(x, k).to(dev), part.shard_rows(g, k).to(dev)
            x_tbl = shard.halo_in.emulate(x_own, xd)
            g_tbl = shard.halo_out.emulate(g_own, gd)
            agg = backend.aggregate(shard.g_in, x_tbl)
            outs.append(backend.transform_fwd(agg, x_own, w, root, bias, False, shard.g_in, table=x_tbl))
            gxs.append(backend.transform_bwd_input(backend.aggregate(shard.g_out, g_tbl), g_own, w, root, None,
                                                   shard.g_out, table=g_tbl))
            a, b, c = backend.transform_bwd_params(agg, x_own, g_own, r, True, True, shard.g_in)
            gw, groot, gbias = gw + a, groot + b, gbias + c
        out_p, gx_p = part.unshard_rows(torch.cat(outs)), part.unshard_rows(torch.cat(gxs))
        if precision == "fp32":
            assert torch.equal(out_p, out1), "forward bits"
            assert torch.equal(gx_p, gx1), "input-gradient bits"
        else:
            for got, want, what in ((out_p, out1, "forward"), (gx_p, gx1, "input gradient")):
                err = float((got - want).abs().max()) / max(1e-30, float(want.abs().max()))
                worst = max(worst, err)
                assert err <= 2e-6, f"{what} {err:.2e}"
        for got, want, what in ((gw, gw1, "grad_weight"), (groot, groot1, "grad_root"), (gbias, gbias1, "grad_bias")):
            err = float((got - want).abs().max()) / max(1e-30, float(want.abs().max()))
            assert err <= 1e-5, f"{what} {err:.2e}"
        print(f"ok   {label}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"FAIL {label}: {type(exc).__name__}: {exc}", flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s; worst split-arithmetic distance from the single-GPU result {worst:.2e}")

#!/usr/bin/env python3
"""Benchmark of the hot path: edges/sec per R-GCN layer (fwd+bwd) on the PrimeKG-shaped
synthetic graph (BASELINE.json metric; configs[1] = C2: 30,926 nodes / 849,456 edges /
3 relations, 64 -> 128 -> 128, fp32 in / fp32 out).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4]

A step = one pass of the two-layer encoder over the whole graph, forward + backward
(conv1 -> relu -> conv2, seeded cotangent; dropout p = 0; bucketing excluded - the graph is
static and bucketed once, the one-time cost is reported in `bucket_ms`).
value = L * E * K / t with L = 2 layers.  N > 1: node-partitioned across the ranks with an
exchange per layer and direction (primekg_rgcn_linkprediction_amd/dist.py), one process
per GPU, launched by torch.distributed.run; `--workload c4` is BASELINE configs[3]
(500k nodes / 20M edge columns / 16 relations) and runs at any N.

Rank 0 prints ONE JSON line.  Beside the contract's fields it carries
  roofline         the gather instantiation with the most time per step: algorithmic bytes / live HIP-event time.
                   While the gathered row table is cache resident (C2: 8-16 MB) the bound is the chip's L2-resident
                   indexed-row rate (`bound: "l2"`, peak 16.8 TB/s, MI355X_MICROARCH.md "Indexed rows"): `frac` is
                   against THAT and is <= 1; the ratio to the HBM peak (> 1 there) is kept as
                   `algorithmic_over_hbm_peak`.  At C4's size the table exceeds the caches: `bound: "hbm"`, peak 8 TB/s.
  dominant_kernel  the launch with the most time per step (a transform GEMM at C2), with BOTH its fractions: of the
                   time its operand bytes need at the achievable streaming rate, and of the dense fp16 MFMA peak
  roofline_mfma    the time-dominant dense transform against the fp32 matrix peak (the arithmetic asked for)
  fp32_mfma_ms_per_step   (N = 1, headline) the same step with RGCN_GEMM_PRECISION=fp32 (exact fp32 MFMA products)
  drop_in          (N = 1, headline) the reference's LITERAL call pattern - RGCNConv.forward twice around F.relu and
                   dropout (src/models/rgcn.py:123-128) - eager and as a replayed HIP graph, beside the two-layer node
  secondary        (headline) `c4_1gpu`: a short run of configs[3]'s graph on this GPU, where the gather IS HBM-bound
                   (the >= 40 %-of-HBM-roofline clause of the north star, driver-timed); at N > 1: `c4`, the same
                   graph node-partitioned over the N ranks
  cpu_baseline     (N = 1) the oracle's PyG-equivalent CPU path on this host, thread count swept
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_STREAM_GBS = 6300.0        # same guide: achievable streaming rate (float4 copy 6.29 TB/s)
L2_GATHER_CEILING_GBS = (16800.0, 18800.0)   # same guide: rows gathered from the XCD's L2, chip-wide
MALL_GATHER_GBS = 8600.0       # same guide: 38 MB table, uniformly random rows (Infinity Cache)
F32_MATRIX_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32 (= fp32 vector rate)
F16_MATRIX_PEAK_TF = 2500.0    # dense fp16 MFMA
LAYERS = 2
PMC_FILES = ("r04_pmc_counters.json", "r03_pmc_counters.json", "r02_pmc_counters.json", "r01_pmc_counters.json")
METRIC = "edges/sec per RGCN layer (fwd+bwd), PrimeKG 30.9k nodes/849k edges/3 rels"

WORKLOADS = {
    "c2": {"dims": (64, 128, 128), "bases": None, "graph": "primekg",
           "name": "C2: PrimeKG-shaped synthetic graph"},
    "c3": {"dims": (64, 256, 256), "bases": 4, "graph": "primekg",
           "name": "C3 (secondary): PrimeKG-shaped synthetic graph, num_bases=4"},
    "c4": {"dims": (64, 128, 128), "bases": None, "graph": "uniform", "nodes": 500_000, "edges": 20_000_000,
           "relations": 16, "name": "C4 (BASELINE configs[3]): uniform synthetic graph"},
}
WORKLOADS["c4-1gpu"] = WORKLOADS["c4"]          # round 2's name for the N = 1 run of the same graph


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--edges", type=int, default=None, help="override E of the PrimeKG-shaped graph (default 849,456)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches, never a HIP graph replay")
    ap.add_argument("--graph", action="store_true", help="time the HIP graph replay even if eager calibrates faster")
    ap.add_argument("--no-replica", action="store_true",
                    help="N > 1: skip the batch-replica leg reported beside the node-partitioned number")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline run: skip the fp32-MFMA pass and the short C4 leg (profiling passes use this)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--fp16-gather", action="store_true",
                    help="BASELINE configs[4]: forward gathers read an fp16 copy of the feature table "
                         "(fp32 accumulate); NOT the headline configuration")
    return ap.parse_args()


def gather_bytes(num_edges, segments, d, weighted):
    """Algorithmic bytes of ONE level-0 gather launch that materialises its output
    (SURVEY.md section 8d / DESIGN.md): every edge reads one d-float row + a 4-byte column id
    (+ a 4-byte 1/cnt weight in the transposed form), plus rowptr and cnt once, plus the
    [segments = N*R, d] output written once."""
    b = num_edges * (4 * d + 4) + 4 * (segments + 1) + 4 * segments * d
    b += 4 * num_edges if weighted else 4 * segments
    return b


def gather_compulsory_bytes(num_edges, segments, table_rows, d, weighted):
    """What HBM must move for that launch if every cache were perfect: the row table ONCE, the ids
    (and weights) once, rowptr / cnt once, the output once."""
    b = 4 * table_rows * d + 4 * num_edges + 4 * (segments + 1) + 4 * segments * d
    b += 4 * num_edges if weighted else 4 * segments
    return b


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from a COMMITTED PMC pass (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 correction applied) - counters cannot be read from inside
    this process, so this is a file constant, labelled as such.  -> (bytes | None, source)"""
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as fh:
                table = json.load(fh)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for kname, entry in table.items():
            if kname.replace(" ", "") == kernel.replace(" ", ""):
                return entry.get("hbm_bytes"), f"profiles/{name} (committed rocprofv3 --pmc passes, not measured in this run)"
    return None, "no committed PMC pass for this kernel"


def cpu_baseline(ei, et, n, r, dims, bases, seconds):
    """PyG-equivalent CPU path (restated; torch_geometric unavailable offline): the oracle's
    op-for-op loop path incl. autograd on this host's cores, same graph/seed/step definition.
    Mask/bucketing time is included, as PyG redoes it every call.  The thread count is swept over
    {8, 16, 32, all}: `scatter_add_` stops scaling (and degrades) long before 128 threads; the
    fastest setting is the baseline."""
    from oracle import rgcn_oracle as O
    torch.manual_seed(0)
    emb = torch.nn.init.xavier_uniform_(torch.empty(n, dims[0])).requires_grad_(True)
    convs = [O.RGCNConvRef(dims[0], dims[1], r, num_bases=bases), O.RGCNConvRef(dims[1], dims[2], r, num_bases=bases)]
    cot = torch.randn(n, dims[2])

    def step():
        h = torch.relu(convs[0](emb, ei, et))
        out = convs[1](h, ei, et)
        emb.grad = None
        for c in convs:
            c.zero_grad(set_to_none=True)
        out.backward(cot)

    all_threads = torch.get_num_threads()
    settings = sorted({t for t in (8, 16, 32, all_threads) if t <= all_threads})
    sweep = {}
    try:
        for threads in settings:
            torch.set_num_threads(threads)
            step()
            times = []
            t_end = time.perf_counter() + seconds / len(settings)
            while time.perf_counter() < t_end or len(times) < 2:
                t0 = time.perf_counter()
                step()
                times.append(time.perf_counter() - t0)
            times.sort()
            sweep[threads] = (times[len(times) // 2], len(times))
    finally:
        torch.set_num_threads(all_threads)
    best = min(sweep, key=lambda t: sweep[t][0])
    med, count = sweep[best]
    return {"value": LAYERS * ei.size(1) / med, "unit": "edges/s", "cores": best,
            "kind": "port", "ms_per_step": med * 1e3,
            "thread_sweep_edges_per_s": {str(t): LAYERS * ei.size(1) / m for t, (m, _) in sweep.items()},
            "sample": f"{count} full encoder fwd+bwd steps (median) of the oracle's PyG-equivalent loop path "
                      f"at {best} threads, the fastest of {settings} on this {os.cpu_count()}-cpu host"}


# ------------------------------------------------------------------------------------------------
# building a workload and timing it
# ------------------------------------------------------------------------------------------------
class Run:
    """one workload on this process's GPU (N = 1) or this rank's shard (N > 1): `step()` = one fwd + bwd"""

    def __init__(self, name, args, dev, dist, world, edges=None, fp16=False):
        from primekg_rgcn_linkprediction_amd import RGCNConv, ops, rgcn_encoder2, synth
        wl = WORKLOADS[name]
        self.name, self.wl, self.dev, self.dist, self.world = name, wl, dev, dist, world
        self.dims, self.bases = wl["dims"], wl["bases"]
        if wl["graph"] == "primekg":
            ei, et, n, r = synth.primekg_like(num_edges=edges or synth.PRIMEKG_EDGES, seed=42)
        else:
            ei, et, n, r = synth.uniform_graph(wl["nodes"], wl["edges"], wl["relations"], seed=42)
        self.ei, self.et, self.n, self.r, self.num_edges = ei, et, n, r, ei.size(1)
        dims = self.dims
        torch.manual_seed(0)
        self.emb_cpu = torch.nn.init.xavier_uniform_(torch.empty(n, dims[0]))
        gdt = torch.float16 if fp16 else None
        self.convs = [RGCNConv(dims[0], dims[1], r, num_bases=self.bases, gather_dtype=gdt),
                      RGCNConv(dims[1], dims[2], r, num_bases=self.bases, gather_dtype=gdt)]
        self.cot_cpu = torch.randn(n, dims[2])
        self.summary = None
        if world == 1:
            eid, etd = ei.to(dev), et.to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph = ops.bucket(eid, etd, n, r)
            torch.cuda.synchronize()
            self.bucket_ms = (time.perf_counter() - t0) * 1e3
            emb = self.emb_cpu.to(dev).requires_grad_(True)
            convs = self.convs = [c.to(dev) for c in self.convs]
            cot = self.cot_cpu.to(dev)
            params = [emb] + [p for c in convs for p in c.parameters()]
            self._keep = (eid, etd)

            def step():
                # DrugDiseaseRGCN.forward with dropout inactive: conv1 -> relu -> conv2
                out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
                for p in params:
                    p.grad = None
                out.backward(cot)
            self.step = step

            def drop_in_step():
                # the reference's call pattern LITERALLY (/root/reference/src/models/rgcn.py:123-128, as INTEGRATION.md
                # section 1's import swap leaves it): RGCNConv.forward twice around F.relu and the dropout module
                x = convs[0](emb, eid, etd)
                x = torch.nn.functional.relu(x)
                x = torch.nn.functional.dropout(x, 0.0, True)
                out = convs[1](x, eid, etd)
                for p in params:
                    p.grad = None
                out.backward(cot)
            self.drop_in_step = drop_in_step
            self.parallelism = "1 GPU"
        else:
            from primekg_rgcn_linkprediction_amd import dist as rdist
            backend = os.environ.get("RGCN_BENCH_BACKEND", "nccl")
            t0 = time.perf_counter()
            enc = self.enc = rdist.PartitionedEncoder(ei, et, n, r, self.emb_cpu, self.convs, dev)
            torch.cuda.synchronize()
            self.bucket_ms = (time.perf_counter() - t0) * 1e3
            cot = enc.shard_rows(self.cot_cpu).to(dev)
            self.step = lambda: enc.step(cot)                                     # noqa: E731
            exchange = "RCCL over xGMI" if backend == "nccl" else f"{backend} (host-staged rehearsal, NOT RCCL)"
            self.summary = summ = enc.exchange_summary()
            how = ("halo all-to-all-v of the rows a rank's edges read, interior rows computed while it is in flight"
                   if summ["scheme"] == "pull" else
                   "partial sums of the boundary rows from the source owner: reduce-scatter forward, halo all-to-all-v backward")
            deal = {"clustered": "locality-aware clustered assignment", "deal": "degree-balanced deal"}.get(summ.get("partition"), "given assignment")
            self.parallelism = (f"node-partitioned x{world} ({deal}), scheme '{summ['scheme']}': {how}, per "
                                f"layer and direction, over {exchange}")

    def describe(self, headline):
        d = self.dims
        return (f"{self.wl['name']}, {self.n} nodes / {self.num_edges} edge columns / {self.r} relations, encoder "
                f"{d[0]}->{d[1]}->{d[2]}, 2 layers fwd+bwd, full graph per step, dropout 0"
                + ("" if headline else "  [NOT the headline configuration]"))

    def sync(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, k):
        self.sync()
        t = time.perf_counter()
        for _ in range(k):
            fn()
        self.sync()
        return (time.perf_counter() - t) / k

    def agree(self, value, op):
        """the same number on every rank (launch-mode decisions must not diverge between ranks)"""
        if self.dist is None:
            return value
        t = torch.tensor([float(value)], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=op)
        return t.item()

    def capture(self):
        """ONE HIP graph of the step (RCCL collectives included at N > 1: they capture like kernels)"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.step()
        torch.cuda.current_stream().wait_stream(side)
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph, capture_error_mode="thread_local"):
            self.step()
        for _ in range(3):
            hip_graph.replay()
        torch.cuda.synchronize()
        return hip_graph

    def measure(self, steps, warmup, use_graph, force_graph=False):
        """-> (seconds for `steps` steps: max over ranks, launch mode).  The step is a fixed sequence of launches on
        a static graph: issued eagerly or replayed from ONE captured HIP graph (same kernels and work either way); a
        short untimed calibration picks the faster launch mode on this machine."""
        for _ in range(warmup):
            self.step()
        self.sync()
        run, mode = self.step, "eager"
        if use_graph:
            hip_graph = self.capture()                         # a failure here is a failure of the run: no silent fallback
            t_graph, t_eager = self.timed(hip_graph.replay, 10), self.timed(self.step, 10)
            if force_graph or t_graph < 1.1 * t_eager:         # replay unless eager is clearly faster (10 iterations each: noisy)
                run, mode = hip_graph.replay, "hipGraph replay"
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            elapsed = self.agree(elapsed, self.dist.ReduceOp.MAX)
        return elapsed, mode

    def events(self, event_steps):
        """Per-kernel durations, live, from HIP events on the launch stream: an eager pass of the same step (events
        cannot be read back from inside a captured graph).  At N > 1 every rank runs the pass - the exchanges are
        collective - and rank 0 reports its own shard."""
        from primekg_rgcn_linkprediction_amd import ops
        ops.GATHER_EVENTS, ops.GEMM_EVENTS, ops.FUSED_EVENTS = [], [], []
        try:
            for _ in range(event_steps):
                # a short device-side spin first, so that the host has queued the step's launches
                # before they execute: the events then bracket back-to-back kernels, not launch gaps
                if self.world == 1:
                    torch.cuda._sleep(4_000_000)
                self.step()
            self.sync()
        finally:
            ev = (ops.GATHER_EVENTS, ops.GEMM_EVENTS, ops.FUSED_EVENTS)
            ops.GATHER_EVENTS = ops.GEMM_EVENTS = ops.FUSED_EVENTS = None
        return ev


def event_overhead_us(world):
    """what an empty bracket costs on this stream (two event records, nothing between): reported, not subtracted -
    the profiler's kernel-only durations in profiles/ are shorter by about this much"""
    if world == 1:
        torch.cuda._sleep(2_000_000)
    pairs = []
    for _ in range(20):
        b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.record()
        e.record()
        pairs.append((b, e))
    torch.cuda.synchronize()
    return sum(b.elapsed_time(e) for b, e in pairs) / len(pairs) * 1e3


def kernel_tables(run, events, event_steps):
    """-> (gather + fused kernels, transform calls) as lists of dicts from the three event lists"""
    gather_events, gemm_events, fused_events = events
    kernels = []
    per, shape = {}, {}
    for transposed, d, edges, segments, table_rows, beg, end in gather_events:
        per.setdefault((transposed, d), []).append(beg.elapsed_time(end) * 1e-3)   # seconds
        shape[(transposed, d)] = (edges, segments, table_rows)
    for (transposed, d), ts in sorted(per.items()):
        avg = sum(ts) / len(ts)
        edges, segments, table_rows = shape[(transposed, d)]
        nbytes = gather_bytes(edges, segments, d, transposed)
        comp = gather_compulsory_bytes(edges, segments, table_rows, d, transposed)
        kernels.append({"kernel": f"k_aggregate<{d // 4},{'true' if transposed else 'false'}>", "kind": "gather",
                        "d": d, "transposed": transposed, "launches_per_step": len(ts) // event_steps,
                        "avg_us": avg * 1e6, "bytes": nbytes, "gbs": nbytes / avg / 1e9,
                        "compulsory_hbm_bytes": comp, "table_bytes": 4 * table_rows * d,
                        "total_us_per_step": sum(ts) / event_steps * 1e6})
    # the one-kernel layers (gather into LDS + transform): algorithmic bytes = what the gather and the rows in /
    # out cost - there is no aggregate write + read; "+store": the kept aggregate's one write
    per, shape = {}, {}
    for kind, rows, rels, edges, hub_rows, dk, dn, beg, end in fused_events:
        per.setdefault((kind, dk, dn), []).append(beg.elapsed_time(end) * 1e-3)
        shape[(kind, dk, dn)] = (rows, rels, edges, hub_rows)
    for (kind, dk, dn), ts in sorted(per.items()):
        rows, rels, edges, hub_rows = shape[(kind, dk, dn)]
        avg = sum(ts) / len(ts)
        weighted = kind.startswith("bwd")
        ids = edges * (8 if weighted else 4) + 4 * (rows * rels + 1)
        dense = 4 * rows * (dk + dn) + (4 * rows * dn if kind.endswith("mask") else 0) \
            + (4 * rows * rels * dk if kind.endswith("store") else 0)
        nbytes = edges * 4 * dk + hub_rows * 4 * dk + ids + dense
        comp = 4 * rows * dk + ids + dense                  # the table once instead of once per edge
        kernels.append({"kernel": f"k_layer_fused<{kind}, {dk}->{dn}>", "kind": "fused layer", "d": dk,
                        "transposed": weighted, "launches_per_step": len(ts) // event_steps, "avg_us": avg * 1e6,
                        "bytes": nbytes, "gbs": nbytes / avg / 1e9, "compulsory_hbm_bytes": comp,
                        "table_bytes": 4 * rows * dk, "total_us_per_step": sum(ts) / event_steps * 1e6,
                        "flops": 2.0 * rows * (rels + 1) * dk * dn})
    calls = []
    per = {}
    for kind, m, k, nn, prec, beg, end in gemm_events:
        per.setdefault((kind, m, k, nn, prec), []).append(beg.elapsed_time(end) * 1e-3)
    for (kind, m, k, nn, prec), ts in sorted(per.items()):
        avg = sum(ts) / len(ts)
        flops = 2.0 * m * k * nn
        # operand bytes of the call: NT calls read A [M, K] once and write C [M, N]; the parameter-gradient call
        # (M = (R + 1) d_in columns, K = node rows, N = d_out) reads [agg | x] and g once (weights / slabs: small)
        nbytes = 4.0 * (k * (m + nn) if kind == "bwd_params" else m * (k + nn))
        calls.append({"call": kind, "M": m, "K": k, "N": nn, "arithmetic": prec,
                      "launches_per_step": len(ts) // event_steps, "avg_us": avg * 1e6, "flops": flops,
                      "tflops": flops / avg / 1e12, "operand_bytes": nbytes,
                      "total_us_per_step": sum(ts) / event_steps * 1e6})
    return kernels, calls


def roofline_of(dom, headline_single):
    """the contract's `roofline` object for the dominant gather / fused-layer kernel"""
    traffic, traffic_source = pmc_traffic(dom["kernel"]) if headline_single else (None, "not applicable")
    cache_resident = dom["table_bytes"] <= 200e6          # fits the 256 MiB Infinity Cache beside the streams
    l2_resident = dom["table_bytes"] <= 32e6              # and the 8 x 4 MiB L2s (C2: 7.9 / 15.8 MB)
    if l2_resident:
        bound, peak = "l2", L2_GATHER_CEILING_GBS[0]
        note = ("the gathered row table is L2 / Infinity-Cache resident, so the launch is bounded by the chip's L2-resident "
                "indexed-row rate (guide: 16.8-18.8 TB/s), not by HBM: `frac` = achieved / 16.8 TB/s; "
                "`algorithmic_over_hbm_peak` (> 1 here) is the same rate against the 8 TB/s HBM peak and "
                "`frac_compulsory` what HBM itself has to deliver")
    else:
        bound, peak = "hbm", HBM_PEAK_GBS
        note = ("the gathered row table exceeds the L2s" + (" (Infinity-Cache resident)" if cache_resident else " and the Infinity Cache")
                + ": `frac` is a fraction of the HBM peak")
    return {"bound": bound, "achieved": dom["gbs"], "peak": peak, "unit": "GB/s", "frac": dom["gbs"] / peak,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": dom["kernel"], "avg_us": dom["avg_us"], "algorithmic_bytes_per_launch": dom["bytes"],
            "algorithmic_over_hbm_peak": dom["gbs"] / HBM_PEAK_GBS,
            "compulsory_hbm_bytes": dom["compulsory_hbm_bytes"],
            "frac_compulsory": dom["compulsory_hbm_bytes"] / (dom["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "table_bytes": dom["table_bytes"], "table_cache_resident": cache_resident,
            "l2_indexed_row_ceiling_gbs": list(L2_GATHER_CEILING_GBS), "note": note}


def dominant_kernel_of(kernels, calls):
    """the launch with the most time per step, with its byte-floor and MFMA fractions"""
    best = None
    for k in kernels:
        floor_us = k["bytes"] / (L2_GATHER_CEILING_GBS[0] * 1e9 if k["table_bytes"] <= 32e6 else HBM_STREAM_GBS * 1e9) * 1e6
        cand = {"kernel": k["kernel"], "kind": k["kind"], "avg_us": k["avg_us"], "launches_per_step": k["launches_per_step"],
                "total_us_per_step": k["total_us_per_step"], "bytes": k["bytes"],
                "byte_floor_us": floor_us, "frac_of_byte_floor": floor_us / k["avg_us"],
                "byte_floor_rate": "L2-resident indexed rows 16.8 TB/s" if k["table_bytes"] <= 32e6 else "HBM streaming 6.3 TB/s",
                "mfma_frac": (3.0 * k["flops"] / (k["avg_us"] * 1e-6) / 1e12 / F16_MATRIX_PEAK_TF) if "flops" in k else None}
        if best is None or cand["total_us_per_step"] > best["total_us_per_step"]:
            best = cand
    for c in calls:
        passes = 3 if c["arithmetic"] == "split" else 1
        peak = F16_MATRIX_PEAK_TF if c["arithmetic"] in ("split", "half", "f16") else F32_MATRIX_PEAK_TF
        floor_us = c["operand_bytes"] / (HBM_STREAM_GBS * 1e9) * 1e6
        cand = {"kernel": f"transform {c['call']} [{c['M']} x {c['K']}] x [{c['K']} x {c['N']}] ({c['arithmetic']})",
                "kind": "transform GEMM", "avg_us": c["avg_us"], "launches_per_step": c["launches_per_step"],
                "total_us_per_step": c["total_us_per_step"], "bytes": c["operand_bytes"],
                "byte_floor_us": floor_us, "frac_of_byte_floor": floor_us / c["avg_us"],
                "byte_floor_rate": "operands once at the achievable streaming rate, 6.3 TB/s",
                "executed_tflops": passes * c["tflops"], "mfma_frac": passes * c["tflops"] / peak,
                "mfma_peak_tflops": peak}
        if best is None or cand["total_us_per_step"] > best["total_us_per_step"]:
            best = cand
    return best


def free_run(run):
    """drop everything a workload holds on the device (its closures pin the tensors), then the cached structures"""
    from primekg_rgcn_linkprediction_amd import ops
    run.__dict__.clear()
    ops.clear_graph_cache()
    gc.collect()
    torch.cuda.empty_cache()


def secondary_c4(args, dev, dist, world):
    """a short run of BASELINE configs[3]'s graph: on ONE GPU the table (128 / 256 MB) is not cache resident and the
    gather is HBM-bound; at N > 1 the graph is node-partitioned over the ranks"""
    run = Run("c4", args, dev, dist, world)
    steps, warmup = 10, 3
    elapsed, mode = run.measure(steps, warmup, use_graph=False)
    out = {"workload": run.describe(False), "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3, "value": LAYERS * run.num_edges * steps / elapsed, "unit": "edges/s",
           "launch": mode, "parallelism": run.parallelism, "bucket_ms": run.bucket_ms}
    if world > 1:
        out["exchange_rank0"] = run.summary
    ev_steps = 3
    kernels, calls = kernel_tables(run, run.events(ev_steps), ev_steps)
    if kernels:
        dom = max(kernels, key=lambda k: k["total_us_per_step"])
        out["roofline"] = roofline_of(dom, False)
        gathers = [k for k in kernels if k["kind"] == "gather"]
        if gathers:
            g = max(gathers, key=lambda k: k["total_us_per_step"])
            out["gather_vs_hbm_peak"] = {"kernel": g["kernel"], "avg_us": g["avg_us"], "achieved_gbs": g["gbs"],
                                         "frac_of_hbm_peak": g["gbs"] / HBM_PEAK_GBS, "table_bytes": g["table_bytes"]}
        out["kernels"] = [{k: v for k, v in kk.items() if k in ("kernel", "avg_us", "gbs", "launches_per_step", "total_us_per_step")}
                          for kk in kernels]
    free_run(run)
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if world > 1 and WORKLOADS[args.workload]["bases"] is not None:
        raise SystemExit("N > 1 runs c2 (headline) or c4 (BASELINE configs[3]); basis-decomposed layers are N = 1 only")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the measured path")

    from primekg_rgcn_linkprediction_amd import _lib, ops, synth
    if not os.path.exists(_lib.LIB_PATH):                 # a fresh checkout: the library is a build product
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        for _ in range(600):                              # the other ranks of the node wait for rank 0's build
            if os.path.exists(_lib.LIB_PATH):
                break
            time.sleep(0.5)
    _lib.load()                                           # fail loudly if the HIP library is missing
    # one process per GPU.  (Rehearsal on a 1-GPU box: RGCN_BENCH_BACKEND=gloo lets several ranks
    # share cuda:0 with host-staged exchanges - RCCL refuses two ranks on one device.)
    backend = os.environ.get("RGCN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    is_c4 = args.workload in ("c4", "c4-1gpu")
    run = Run(args.workload, args, dev, dist, world, edges=args.edges, fp16=args.fp16_gather)
    use_graph = world == 1 and not args.no_graph
    elapsed, launch_mode = run.measure(args.steps, args.warmup, use_graph, force_graph=args.graph)

    event_steps = min(args.steps, 20)
    events = run.events(event_steps)
    overhead_us = event_overhead_us(world)

    headline = args.workload == "c2" and not args.fp16_gather and args.edges in (None, synth.PRIMEKG_EDGES)
    result = {
        "metric": METRIC,
        "value": LAYERS * run.num_edges * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32" if not args.fp16_gather else "f16 feature table, f32 accumulate/transform"),
        "data": "synthetic",
        "config": {"workload": run.describe(headline),
                   "parallelism": run.parallelism, "launch": launch_mode,
                   "transform_arithmetic": ("fp32 values as fp16 hi/lo pairs on the fp16 matrix cores, fp32 accumulate "
                                            "(3 MFMA passes; 1e-5 gates of tests/test_gpu_parity.py)"
                                            if ops.GEMM_PRECISION == "split" else "fp32 MFMA")},
        "bucket_ms": run.bucket_ms,
    }
    if world > 1:
        result["exchange_rank0"] = run.summary   # rows received per exchange as a fraction of the rows rank 0 does not own

    kernels, calls = kernel_tables(run, events, event_steps)
    if kernels:
        dom = max(kernels, key=lambda k: k["total_us_per_step"])
        result["roofline"] = roofline_of(dom, world == 1 and headline)
        result["roofline"]["event_bracket_overhead_us"] = overhead_us
        result["gather_kernels"] = kernels
    if calls:
        domg = max(calls, key=lambda c: c["total_us_per_step"])
        executed = domg["flops"] * (3 if domg["arithmetic"] == "split" else 1)
        result["roofline_mfma"] = {
            "bound": "mfma", "call": f"{domg['call']} [{domg['M']} x {domg['K']}] x [{domg['K']} x {domg['N']}]",
            "arithmetic": domg["arithmetic"], "avg_us": domg["avg_us"],
            "achieved": domg["tflops"], "peak": F32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
            "frac": domg["tflops"] / F32_MATRIX_PEAK_TF,
            "executed_tflops": executed / (domg["avg_us"] * 1e-6) / 1e12,
            "executed_peak": F16_MATRIX_PEAK_TF if domg["arithmetic"] in ("split", "f16") else F32_MATRIX_PEAK_TF,
            # what a register-fed loop of that instruction sustains on this chip (tools/mfma_f16_peak.hip, a committed
            # measurement, not part of this run): 20.5 ns per v_mfma_f32_32x32x16_f16 and SIMD, ~1.56 GHz under matrix load
            "executed_peak_sustained_measured": 1640.0 if domg["arithmetic"] in ("split", "f16") else None,
            "executed_peak_sustained_source": "profiles/r03_mfma_f16_peak.txt",
            "note": ("flops = 2 M K N of the fp32 contraction the caller asked for, against the fp32 matrix peak; "
                     "in split precision the call executes 3 fp16 MFMA passes (`executed_tflops`, against the dense fp16 "
                     "peak) and its bracket includes the operand scan and the weight split launches"),
            "sum_transform_us_per_step": sum(c["total_us_per_step"] for c in calls)}
        result["transform_calls"] = calls
    if kernels or calls:
        result["dominant_kernel"] = dominant_kernel_of(kernels, calls)

    if world > 1 and not args.no_replica and not is_c4:
        # Reported beside the node-partitioned number (never instead of it): batch-replica mode,
        # every GPU the whole graph and its own mini-batch, one flat all-reduce of all parameter
        # gradients per step.  Per-GPU work is fixed, so this one is weak scaling.
        import copy
        from primekg_rgcn_linkprediction_amd import dist as rdist
        rep = rdist.ReplicatedEncoder(run.ei, run.et, run.n, run.r, run.emb_cpu, [copy.deepcopy(c) for c in run.convs], dev)
        cot_full = run.cot_cpu.to(dev)
        for _ in range(args.warmup):
            rep.step(cot_full)
        rep_step = lambda: rep.step(cot_full)                              # noqa: E731
        rep_s = run.agree(run.timed(rep_step, args.steps), dist.ReduceOp.MAX)
        result["replica"] = {"value": world * LAYERS * run.num_edges / rep_s, "unit": "edges/s",
                             "ms_per_step": rep_s * 1e3, "scaling": "weak", "launch": "eager",
                             "parallelism": f"batch replicas x{world}: full graph and encoder per GPU, one "
                                            f"{rep._flat.numel() * 4 / 1e6:.1f} MB gradient all-reduce per step"}
        del rep, cot_full

    exit_code = 0
    if world > 1:
        # The numbers above are eager launches and are what is reported.  Capturing the N > 1 step -
        # collectives included - into one HIP graph is OPT-IN (RGCN_BENCH_GRAPH_N=1): a capture that
        # fails leaves HIP unusable for the rest of the process and a collective inside a replay can
        # hang, so the attempt comes last, its outcome is written into the line, and the process exits
        # non-zero if it failed or hung (the eager line is still printed first).
        result["graph_attempt"] = "not attempted (opt in with RGCN_BENCH_GRAPH_N=1)"

    # ---- secondary legs of the headline line (short; the whole command stays well inside the driver's limit) ----
    want_graph_n = world > 1 and os.environ.get("RGCN_BENCH_GRAPH_N", "0") == "1" and not args.no_graph
    if headline and not args.no_secondary and want_graph_n:
        result["secondary"] = {"skipped": "RGCN_BENCH_GRAPH_N=1: the graph attempt may end the process, so it runs alone"}
    if headline and not args.no_secondary and not want_graph_n:
        secondary = {}
        if world == 1:
            # the same step with exact fp32 MFMA products (RGCN_GEMM_PRECISION=fp32): every module reads
            # ops.GEMM_PRECISION at call time, so the switch is in-process; same graph, same tensors
            saved = ops.GEMM_PRECISION
            try:
                ops.GEMM_PRECISION = "fp32"
                fp32_s, fp32_mode = run.measure(20, 5, use_graph, force_graph=args.graph)
                result["fp32_mfma_ms_per_step"] = fp32_s / 20 * 1e3
                result["fp32_mfma"] = {"ms_per_step": fp32_s / 20 * 1e3, "value": LAYERS * run.num_edges * 20 / fp32_s,
                                       "unit": "edges/s", "steps": 20, "launch": fp32_mode,
                                       "arithmetic": "v_mfma_f32_32x32x2_f32: every product exact fp32 (bit for bit an fmaf chain)"}
            finally:
                ops.GEMM_PRECISION = saved
        if world == 1:
            # the literal drop-in: what a reference user gets from the import swap alone - conv1 -> F.relu -> dropout
            # -> conv2 through RGCNConv.forward, each layer its own autograd node (its own scale launch, no ReLU in a
            # GEMM epilogue) - eager and as one replayed HIP graph, beside the headline's two-layer node
            try:
                headline_step, run.step = run.step, run.drop_in_step
                eager_s, _ = run.measure(20, 8, False)
                graph_s, graph_mode = run.measure(20, 3, use_graph, force_graph=True)
                result["drop_in"] = {"ms_per_step_eager": eager_s / 20 * 1e3, "ms_per_step_graph": graph_s / 20 * 1e3,
                                     "launch_graph": graph_mode, "steps": 20,
                                     "value_graph": LAYERS * run.num_edges * 20 / graph_s, "unit": "edges/s",
                                     "vs_headline": (graph_s / 20 * 1e3) / result["ms_per_step"],
                                     "call_pattern": "conv2(dropout(relu(conv1(x, ei, et)), p=0), ei, et) through RGCNConv.forward "
                                                     "(reference src/models/rgcn.py:123-128), out.backward(cotangent)"}
            except Exception as exc:                                       # the headline line must survive this leg
                result["drop_in"] = {"error": repr(exc)}
                exit_code = exit_code or 6
            finally:
                run.step = headline_step
        keep_cpu = (run.ei, run.et, run.n, run.r, run.dims, run.bases)
        free_run(run)
        run = None
        try:
            secondary["c4_1gpu" if world == 1 else "c4"] = secondary_c4(args, dev, dist, world)
        except Exception as exc:                                           # the headline line must survive this leg
            secondary["c4_1gpu" if world == 1 else "c4"] = {"error": repr(exc)}
            exit_code = exit_code or 5
        result["secondary"] = secondary
    else:
        keep_cpu = (run.ei, run.et, run.n, run.r, run.dims, run.bases)

    if want_graph_n and run is not None:
        import threading

        def bail_out():                                                    # pragma: no cover
            result["graph_attempt"] = "hung: a replayed collective never completed; eager numbers reported"
            if rank == 0:
                print(json.dumps(result), flush=True)
            os._exit(4)

        watchdog = threading.Timer(float(os.environ.get("RGCN_BENCH_GRAPH_TIMEOUT", "120")), bail_out)
        watchdog.daemon = True
        watchdog.start()
        try:
            g = run.capture()
            g_s = run.agree(run.timed(g.replay, args.steps), dist.ReduceOp.MAX)
            result["graph_attempt"] = "ok"
            result["graph_ms_per_step"] = g_s * 1e3
            if g_s < result["ms_per_step"] * 1e-3:
                result["eager_ms_per_step"] = result["ms_per_step"]
                result.update(value=LAYERS * run.num_edges / g_s, ms_per_step=g_s * 1e3)
                result["config"]["launch"] = "hipGraph replay"
        except Exception as exc:                                           # pragma: no cover
            result["graph_attempt"] = f"failed: {exc!r}; eager numbers reported"
            watchdog.cancel()
            if rank == 0:
                print(json.dumps(result), flush=True)
            sys.stderr.flush()
            os._exit(3)                       # HIP is unusable in this process now: no teardown, non-zero exit
        finally:
            watchdog.cancel()

    if rank == 0 and world == 1 and is_c4 and not args.no_cpu_baseline:
        result["cpu_baseline"] = None            # one oracle step over 20M edge columns takes minutes: timed for C2 only
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        ei, et, n, r, dims, bases = keep_cpu
        result["cpu_baseline"] = cpu_baseline(ei, et, n, r, dims, bases, args.cpu_seconds)
        result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:                                                  # pragma: no cover
            pass
    sys.exit(exit_code)


if __name__ == "__main__":
    main()

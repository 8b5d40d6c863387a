#!/usr/bin/env python3
"""Benchmark of the hot path: edges/sec per R-GCN layer (fwd+bwd) on the PrimeKG-shaped
synthetic graph (BASELINE.json metric; configs[1] = C2: 30,926 nodes / 849,456 edges /
3 relations, 64 -> 128 -> 128, fp32 in / fp32 out).

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4-1gpu]

A step = one pass of the two-layer encoder over the whole graph, forward + backward
(conv1 -> relu -> conv2, seeded cotangent; dropout p = 0; bucketing excluded - the graph is
static and bucketed once, the one-time cost is reported in `bucket_ms`).
value = L * E * K / t with L = 2 layers.  N > 1: node-partitioned across the ranks with an
exchange per layer and direction (primekg_rgcn_linkprediction_amd/dist.py), one process
per GPU, launched by torch.distributed.run.

Rank 0 prints ONE JSON line.  Beside the contract's fields it carries
  roofline       the dominant gather kernel: algorithmic bytes / live HIP-event time against the HBM peak
                 (the contract's figure; exceeds 1 at C2 because the 8-16 MB row table is L2/Infinity-Cache
                 resident) AND the two figures that mean something there: compulsory HBM bytes / time
                 against the HBM peak, and the achieved rate against the chip's measured L2-resident
                 indexed-row ceiling (MI355X_MICROARCH.md, "Indexed rows: gather into LDS")
  roofline_mfma  the time-dominant dense transform: flops / live HIP-event time against the fp32 matrix
                 peak (the arithmetic the caller asked for) and, in split precision, the executed fp16
                 MFMA flops (3 passes) against the fp16 matrix peak
  cpu_baseline   (N = 1) the oracle's PyG-equivalent CPU path on this host, thread count swept
The secondary workloads (`--workload c3`, `--workload c4-1gpu`: C4's 500k-node / 20M-edge / 16-relation
graph on ONE GPU, where the 256 MB table is no longer cache resident) print the same line for their
shape; the headline stays C2.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
L2_GATHER_CEILING_GBS = (16800.0, 18800.0)   # same guide: rows gathered from the XCD's L2, chip-wide
MALL_GATHER_GBS = 8600.0       # same guide: 38 MB table, uniformly random rows (Infinity Cache)
F32_MATRIX_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32 (= fp32 vector rate)
F16_MATRIX_PEAK_TF = 2500.0    # dense fp16 MFMA
LAYERS = 2
PMC_FILES = ("r02_pmc_counters.json", "r01_pmc_counters.json")

WORKLOADS = {
    "c2": {"dims": (64, 128, 128), "bases": None, "graph": "primekg",
           "name": "C2: PrimeKG-shaped synthetic graph"},
    "c3": {"dims": (64, 256, 256), "bases": 4, "graph": "primekg",
           "name": "C3 (secondary): PrimeKG-shaped synthetic graph, num_bases=4"},
    "c4-1gpu": {"dims": (64, 128, 128), "bases": None, "graph": "uniform", "nodes": 500_000, "edges": 20_000_000,
                "relations": 16, "name": "C4 on ONE GPU (secondary): uniform synthetic graph"},
}


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
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--fp16-gather", action="store_true",
                    help="BASELINE configs[4]: forward gathers read an fp16 copy of the feature table "
                         "(fp32 accumulate); NOT the headline configuration")
    return ap.parse_args()


def gather_bytes(num_edges, num_nodes, num_relations, d, weighted):
    """Algorithmic bytes of ONE level-0 gather launch that materialises its output
    (SURVEY.md section 8d / DESIGN.md): every edge reads one d-float row + a 4-byte column id
    (+ a 4-byte 1/cnt weight in the transposed form), plus rowptr and cnt once, plus the
    [N*R, d] output written once."""
    nr = num_nodes * num_relations
    b = num_edges * (4 * d + 4) + 4 * (nr + 1) + 4 * nr * d
    b += 4 * num_edges if weighted else 4 * nr
    return b


def gather_compulsory_bytes(num_edges, num_nodes, num_relations, d, weighted):
    """What HBM must move for that launch if every cache were perfect: the row table ONCE, the ids
    (and weights) once, rowptr / cnt once, the output once."""
    nr = num_nodes * num_relations
    b = 4 * num_nodes * d + 4 * num_edges + 4 * (nr + 1) + 4 * nr * d
    b += 4 * num_edges if weighted else 4 * nr
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
    if world > 1 and args.workload != "c2":
        raise SystemExit("N > 1 runs the headline workload (c2) only")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the measured path")

    from primekg_rgcn_linkprediction_amd import RGCNConv, _lib, ops, rgcn_encoder2, synth
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

    wl = WORKLOADS[args.workload]
    dims, bases = wl["dims"], wl["bases"]
    if wl["graph"] == "primekg":
        ei, et, n, r = synth.primekg_like(num_edges=args.edges or synth.PRIMEKG_EDGES, seed=42)
    else:
        ei, et, n, r = synth.uniform_graph(wl["nodes"], wl["edges"], wl["relations"], seed=42)
    num_edges = ei.size(1)
    torch.manual_seed(0)
    emb_cpu = torch.nn.init.xavier_uniform_(torch.empty(n, dims[0]))
    gdt = torch.float16 if args.fp16_gather else None
    convs = [RGCNConv(dims[0], dims[1], r, num_bases=bases, gather_dtype=gdt),
             RGCNConv(dims[1], dims[2], r, num_bases=bases, gather_dtype=gdt)]
    cot_cpu = torch.randn(n, dims[2])

    if world == 1:
        eid, etd = ei.to(dev), et.to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.bucket(eid, etd, n, r)
        torch.cuda.synchronize()
        bucket_ms = (time.perf_counter() - t0) * 1e3
        emb = emb_cpu.to(dev).requires_grad_(True)
        convs = [c.to(dev) for c in convs]
        cot = cot_cpu.to(dev)
        params = [emb] + [p for c in convs for p in c.parameters()]

        def step():
            # DrugDiseaseRGCN.forward with dropout inactive: conv1 -> relu -> conv2
            out = rgcn_encoder2(emb, eid, etd, convs[0], convs[1])
            for p in params:
                p.grad = None
            out.backward(cot)
        parallelism = "1 GPU"
    else:
        from primekg_rgcn_linkprediction_amd import dist as rdist
        t0 = time.perf_counter()
        enc = rdist.PartitionedEncoder(ei, et, n, r, emb_cpu, convs, dev)
        torch.cuda.synchronize()
        bucket_ms = (time.perf_counter() - t0) * 1e3
        cot = enc.shard_rows(cot_cpu).to(dev)
        step = lambda: enc.step(cot)                                     # noqa: E731
        exchange = "RCCL over xGMI" if backend == "nccl" else f"{backend} (host-staged rehearsal, NOT RCCL)"
        summ = enc.exchange_summary()
        how = ("halo all-to-all-v of the rows a rank's edges read" if summ["scheme"] == "pull" else
               "partial sums from the source owner: reduce-scatter forward, halo all-to-all-v backward")
        parallelism = (f"node-partitioned x{world} (degree-balanced deal), scheme '{summ['scheme']}': {how}, per layer "
                       f"and direction, over {exchange}")

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()

    # The step is a fixed sequence of launches on a static graph.  It can be issued eagerly (the
    # host runs ahead of the GPU) or replayed from ONE captured HIP graph (no per-launch host
    # cost, robust against a busy host): same kernels and work either way.  A short untimed
    # calibration picks the faster launch mode on this machine; --no-graph / --graph force one.
    def timed(fn, k):
        sync()
        t = time.perf_counter()
        for _ in range(k):
            fn()
        sync()
        return (time.perf_counter() - t) / k

    def agree(value, op):
        """the same number on every rank (launch-mode decisions must not diverge between ranks)"""
        if dist is None:
            return value
        t = torch.tensor([float(value)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return t.item()

    def capture(step_fn):
        """ONE HIP graph of the step (RCCL collectives included at N > 1: they capture like kernels)"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_fn()
        torch.cuda.current_stream().wait_stream(side)
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph, capture_error_mode="thread_local"):
            step_fn()
        for _ in range(3):
            hip_graph.replay()
        torch.cuda.synchronize()
        return hip_graph

    run, launch_mode = step, "eager"
    if world == 1 and not args.no_graph:
        hip_graph = capture(step)                          # a failure here is a failure of the run: no silent fallback
        t_graph, t_eager = timed(hip_graph.replay, 10), timed(step, 10)
        if args.graph or t_graph < 1.1 * t_eager:         # replay unless eager is clearly faster (10 iterations each: noisy)
            run, launch_mode = hip_graph.replay, "hipGraph replay"
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    elapsed = time.perf_counter() - t0

    # Per-kernel durations, live, from HIP events on the launch stream: an eager pass of the same
    # step (events cannot be read back from inside a captured graph).  (At N > 1 every rank runs
    # the pass - the exchanges are collective - and rank 0 reports its own shard.)
    event_steps = min(args.steps, 20)
    ops.GATHER_EVENTS, ops.GEMM_EVENTS, ops.FUSED_EVENTS = [], [], []
    for _ in range(event_steps):
        # a short device-side spin first, so that the host has queued the step's launches
        # before they execute: the events then bracket back-to-back kernels, not launch gaps
        if world == 1:
            torch.cuda._sleep(4_000_000)
        step()
    sync()
    events, ops.GATHER_EVENTS = ops.GATHER_EVENTS, None
    gemm_events, ops.GEMM_EVENTS = ops.GEMM_EVENTS, None
    fused_events, ops.FUSED_EVENTS = ops.FUSED_EVENTS, None
    # what an empty bracket costs on this stream (two event records, nothing between): reported, not
    # subtracted - the profiler's kernel-only durations in profiles/ are shorter by about this much
    if world == 1:
        torch.cuda._sleep(2_000_000)
    pairs = []
    for _ in range(20):
        b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.record()
        e.record()
        pairs.append((b, e))
    torch.cuda.synchronize()
    event_overhead_us = sum(b.elapsed_time(e) for b, e in pairs) / len(pairs) * 1e3
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    headline = args.workload == "c2" and not args.fp16_gather and args.edges in (None, synth.PRIMEKG_EDGES)
    result = {
        "metric": "edges/sec per RGCN layer (fwd+bwd), PrimeKG 30.9k nodes/849k edges/3 rels",
        "value": LAYERS * num_edges * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": ("f32" if not args.fp16_gather else "f16 feature table, f32 accumulate/transform"),
        "data": "synthetic",
        "config": {"workload": f"{wl['name']}, {n} nodes / {num_edges} edge columns / {r} relations, encoder "
                               f"{dims[0]}->{dims[1]}->{dims[2]}, 2 layers fwd+bwd, full graph per step, dropout 0"
                               + ("" if headline else "  [NOT the headline configuration]"),
                   "parallelism": parallelism, "launch": launch_mode,
                   "transform_arithmetic": ("fp32 values as fp16 hi/lo pairs on the fp16 matrix cores, fp32 accumulate "
                                            "(3 MFMA passes; 1e-5 gates of tests/test_gpu_parity.py)"
                                            if ops.GEMM_PRECISION == "split" else "fp32 MFMA")},
        "bucket_ms": bucket_ms,
    }
    if world > 1:
        result["exchange_rank0"] = summ          # rows received per exchange as a fraction of the rows rank 0 does not own

    fused_kernels = []
    if fused_events:
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
            fused_kernels.append({"kernel": f"k_layer_fused<{kind}, {dk}->{dn}>", "d": dk, "transposed": weighted,
                                  "launches_per_step": len(ts) // event_steps, "avg_us": avg * 1e6, "bytes": nbytes,
                                  "gbs": nbytes / avg / 1e9, "compulsory_hbm_bytes": comp, "table_bytes": 4 * rows * dk,
                                  "total_us_per_step": sum(ts) / event_steps * 1e6,
                                  "flops": 2.0 * rows * (rels + 1) * dk * dn})

    if events or fused_kernels:
        # per instantiation of the gather kernel: average duration from the live HIP events
        per, shape = {}, {}
        for transposed, d, edges, segments, beg, end in events:
            per.setdefault((transposed, d), []).append(beg.elapsed_time(end) * 1e-3)   # seconds
            shape[(transposed, d)] = (edges, segments)
        kernels = []
        for (transposed, d), ts in sorted(per.items()):
            avg = sum(ts) / len(ts)
            edges, segments = shape[(transposed, d)]
            nbytes = gather_bytes(edges, segments // r, r, d, transposed)
            comp = gather_compulsory_bytes(edges, segments // r, r, d, transposed)
            kernels.append({"kernel": f"k_aggregate<{d // 4},{'true' if transposed else 'false'}>",
                            "d": d, "transposed": transposed, "launches_per_step": len(ts) // event_steps,
                            "avg_us": avg * 1e6, "bytes": nbytes, "gbs": nbytes / avg / 1e9,
                            "compulsory_hbm_bytes": comp, "table_bytes": 4 * (segments // r) * d,
                            "total_us_per_step": sum(ts) / event_steps * 1e6})
        kernels += fused_kernels
        dom = max(kernels, key=lambda k: k["total_us_per_step"])
        traffic, traffic_source = pmc_traffic(dom["kernel"]) if (world == 1 and headline) else (None, "not applicable")
        cache_resident = dom["table_bytes"] <= 200e6          # fits the 256 MiB Infinity Cache beside the streams
        result["roofline"] = {
            "bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["gbs"] / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": dom["kernel"], "avg_us": dom["avg_us"], "algorithmic_bytes_per_launch": dom["bytes"],
            "compulsory_hbm_bytes": dom["compulsory_hbm_bytes"],
            "frac_compulsory": dom["compulsory_hbm_bytes"] / (dom["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "table_bytes": dom["table_bytes"], "table_cache_resident": cache_resident,
            "l2_indexed_row_ceiling_gbs": list(L2_GATHER_CEILING_GBS),
            "frac_of_l2_ceiling": [dom["gbs"] / L2_GATHER_CEILING_GBS[1], dom["gbs"] / L2_GATHER_CEILING_GBS[0]],
            "note": ("the gathered row table is L2 / Infinity-Cache resident: `frac` (algorithmic bytes against the HBM "
                     "peak, the contract's figure) can exceed 1; `frac_of_l2_ceiling` is the meaningful fraction here and "
                     "`frac_compulsory` is what HBM itself has to deliver"
                     if cache_resident else
                     "the gathered row table exceeds the caches: `frac` is a true fraction of the HBM peak"),
            "event_bracket_overhead_us": event_overhead_us}
        result["gather_kernels"] = kernels

    if gemm_events:
        per = {}
        for kind, m, k, nn, prec, beg, end in gemm_events:
            per.setdefault((kind, m, k, nn, prec), []).append(beg.elapsed_time(end) * 1e-3)
        calls = []
        for (kind, m, k, nn, prec), ts in sorted(per.items()):
            avg = sum(ts) / len(ts)
            flops = 2.0 * m * k * nn
            calls.append({"call": kind, "M": m, "K": k, "N": nn, "arithmetic": prec,
                          "launches_per_step": len(ts) // event_steps, "avg_us": avg * 1e6, "flops": flops,
                          "tflops": flops / avg / 1e12, "total_us_per_step": sum(ts) / event_steps * 1e6})
        domg = max(calls, key=lambda c: c["total_us_per_step"])
        executed = domg["flops"] * (3 if domg["arithmetic"] == "split" else 1)
        result["roofline_mfma"] = {
            "bound": "mfma", "call": f"{domg['call']} [{domg['M']} x {domg['K']}] x [{domg['K']} x {domg['N']}]",
            "arithmetic": domg["arithmetic"], "avg_us": domg["avg_us"],
            "achieved": domg["tflops"], "peak": F32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
            "frac": domg["tflops"] / F32_MATRIX_PEAK_TF,
            "executed_tflops": executed / (domg["avg_us"] * 1e-6) / 1e12,
            "executed_peak": F16_MATRIX_PEAK_TF if domg["arithmetic"] in ("split", "f16") else F32_MATRIX_PEAK_TF,
            "note": ("flops = 2 M K N of the fp32 contraction the caller asked for, against the fp32 matrix peak; "
                     "in split precision the call executes 3 fp16 MFMA passes (`executed_tflops`, against the dense fp16 "
                     "peak) and its bracket includes the operand scan and the weight split launches"),
            "sum_transform_us_per_step": sum(c["total_us_per_step"] for c in calls)}
        result["transform_calls"] = calls

    if world > 1 and not args.no_replica:
        # Reported beside the node-partitioned number (never instead of it): batch-replica mode,
        # every GPU the whole graph and its own mini-batch, one flat all-reduce of all parameter
        # gradients per step.  Per-GPU work is fixed, so this one is weak scaling.
        import copy
        rep = rdist.ReplicatedEncoder(ei, et, n, r, emb_cpu, [copy.deepcopy(c) for c in convs], dev)
        cot_full = cot_cpu.to(dev)
        for _ in range(args.warmup):
            rep.step(cot_full)
        rep_step = lambda: rep.step(cot_full)                              # noqa: E731
        rep_s = agree(timed(rep_step, args.steps), dist.ReduceOp.MAX)
        result["replica"] = {"value": world * LAYERS * num_edges / rep_s, "unit": "edges/s",
                             "ms_per_step": rep_s * 1e3, "scaling": "weak", "launch": "eager",
                             "parallelism": f"batch replicas x{world}: full graph and encoder per GPU, one "
                                            f"{rep._flat.numel() * 4 / 1e6:.1f} MB gradient all-reduce per step"}

    exit_code = 0
    if world > 1:
        # The numbers above are eager launches and are what is reported.  Capturing the N > 1 step -
        # collectives included - into one HIP graph is OPT-IN (RGCN_BENCH_GRAPH_N=1): a capture that
        # fails leaves HIP unusable for the rest of the process and a collective inside a replay can
        # hang, so the attempt comes last, its outcome is written into the line, and the process exits
        # non-zero if it failed or hung (the eager line is still printed first).
        result["graph_attempt"] = "not attempted (opt in with RGCN_BENCH_GRAPH_N=1)"
        if os.environ.get("RGCN_BENCH_GRAPH_N", "0") == "1" and not args.no_graph:
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
                g = capture(step)
                g_s = agree(timed(g.replay, args.steps), dist.ReduceOp.MAX)
                result["graph_attempt"] = "ok"
                result["graph_ms_per_step"] = g_s * 1e3
                if g_s < result["ms_per_step"] * 1e-3:
                    result["eager_ms_per_step"] = result["ms_per_step"]
                    result.update(value=LAYERS * num_edges / g_s, ms_per_step=g_s * 1e3)
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

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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

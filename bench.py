#!/usr/bin/env python3
"""Benchmark of the hot path: edges/sec per R-GCN layer (fwd+bwd) on the PrimeKG-shaped
synthetic graph (BASELINE.json metric; configs[1] = C2: 30,926 nodes / 849,456 edges /
3 relations, 64 -> 128 -> 128, fp32).

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the two-layer encoder over the whole graph, forward + backward
(conv1 -> relu -> conv2, seeded cotangent; dropout p = 0; bucketing excluded - the graph is
static and bucketed once, the one-time cost is reported in `bucket_ms`).
value = L * E * K / t with L = 2 layers.  N > 1: node-partitioned across the ranks with an
RCCL exchange per layer and direction (primekg_rgcn_linkprediction_amd/dist.py), one process
per GPU, launched by torch.distributed.run.

Rank 0 prints ONE JSON line, carrying `roofline` (the dominant gather kernel, HIP events
recorded live inside the timed region on the launch stream) and, at N = 1, `cpu_baseline`
(the PyG-equivalent CPU path of oracle/ timed on this host's cores on a bounded sample).
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
DIMS = (64, 128, 128)
LAYERS = 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--edges", type=int, default=None, help="override E (default 849,456)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches, never a HIP graph replay")
    ap.add_argument("--graph", action="store_true", help="time the HIP graph replay even if eager calibrates faster")
    ap.add_argument("--no-replica", action="store_true",
                    help="N > 1: skip the batch-replica leg reported beside the node-partitioned number")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 correction applied: profiles/r01_pmc_counters.json);
    None when the summary is absent.  Counters cannot be read from inside this process."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_counters.json")
    try:
        with open(path) as fh:
            table = json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    for name, entry in table.items():
        if name.replace(" ", "") == kernel.replace(" ", ""):
            return entry.get("hbm_bytes")
    return None


def cpu_baseline(ei, et, n, r, seconds):
    """PyG-equivalent CPU path (restated; torch_geometric unavailable offline): the oracle's
    op-for-op loop path incl. autograd, all host cores, same graph/seed/step definition.
    Mask/bucketing time is included, as PyG redoes it every call."""
    from oracle import rgcn_oracle as O
    torch.manual_seed(0)
    emb = torch.nn.init.xavier_uniform_(torch.empty(n, DIMS[0])).requires_grad_(True)
    convs = [O.RGCNConvRef(DIMS[0], DIMS[1], r), O.RGCNConvRef(DIMS[1], DIMS[2], r)]
    cot = torch.randn(n, DIMS[2])

    def step():
        h = torch.relu(convs[0](emb, ei, et))
        out = convs[1](h, ei, et)
        emb.grad = None
        for c in convs:
            c.zero_grad(set_to_none=True)
        out.backward(cot)

    step()
    times = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(times) < 3:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": LAYERS * ei.size(1) / med, "unit": "edges/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": med * 1e3,
            "sample": f"{len(times)} full C2 encoder fwd+bwd steps (median) of the oracle's "
                      f"PyG-equivalent loop path, {torch.get_num_threads()} threads"}


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

    ei, et, n, r = synth.primekg_like(num_edges=args.edges or synth.PRIMEKG_EDGES, seed=42)
    num_edges = ei.size(1)
    torch.manual_seed(0)
    emb_cpu = torch.nn.init.xavier_uniform_(torch.empty(n, DIMS[0]))
    gdt = torch.float16 if args.fp16_gather else None
    convs = [RGCNConv(DIMS[0], DIMS[1], r, gather_dtype=gdt), RGCNConv(DIMS[1], DIMS[2], r, gather_dtype=gdt)]
    cot_cpu = torch.randn(n, DIMS[2])

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
        parallelism = f"node-partitioned x{world} (edge-balanced ranges), RCCL exchange per layer"

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

    def pick_launch_mode(step_fn):
        """eager launches or replay of ONE captured HIP graph of the same step (RCCL collectives
        included at N > 1: they capture like kernels), whichever calibrates faster here."""
        allow = not args.no_graph and (world == 1 or os.environ.get("RGCN_BENCH_GRAPH_N", "1") != "0")
        if not allow:
            return step_fn, "eager"
        hip_graph, ok = None, 1.0
        try:
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
        except Exception as exc:                                   # pragma: no cover
            print(f"bench: HIP graph capture failed ({exc!r}); timing eager launches", file=sys.stderr)
            ok = 0.0
        if agree(ok, dist.ReduceOp.MIN if dist is not None else None) < 1.0:
            return step_fn, "eager"
        t_graph = agree(timed(hip_graph.replay, 10), dist.ReduceOp.MAX if dist is not None else None)
        t_eager = agree(timed(step_fn, 10), dist.ReduceOp.MAX if dist is not None else None)
        if args.graph or t_graph < t_eager:
            return hip_graph.replay, "hipGraph replay"
        return step_fn, "eager"

    # N = 1: calibrate now.  N > 1: time the eager launches first - that result is safe whatever
    # happens later - and try the captured graph (RCCL collectives included) at the very end: a capture
    # that fails leaves HIP unusable for the rest of the process, so nothing may depend on it.
    run, launch_mode = pick_launch_mode(step) if world == 1 else (step, "eager")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    elapsed = time.perf_counter() - t0

    # Per-kernel durations of the gather, live, from HIP events on the launch stream: an eager
    # pass of the same step (events cannot be read back from inside a captured graph).
    # (At N > 1 every rank runs the pass - the exchanges are collective - and rank 0 reports the
    # gather over its own shard, whose edge count sizes the algorithmic bytes.)
    event_steps = min(args.steps, 20)
    ops.GATHER_EVENTS = []
    for _ in range(event_steps):
        # a short device-side spin first, so that the host has queued the step's launches
        # before they execute: the events then bracket back-to-back kernels, not launch gaps
        if world == 1:
            torch.cuda._sleep(2_000_000)
        step()
    sync()
    events, ops.GATHER_EVENTS = ops.GATHER_EVENTS, None
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

    result = {
        "metric": "edges/sec per RGCN layer (fwd+bwd), PrimeKG 30.9k nodes/849k edges/3 rels",
        "value": LAYERS * num_edges * args.steps / elapsed,
        "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if not args.fp16_gather else "f16 feature table, f32 accumulate/transform",
        "data": "synthetic",
        "config": {"workload": f"C2: PrimeKG-shaped synthetic graph, {n} nodes / {num_edges} edge columns / "
                               f"{r} relations, encoder {DIMS[0]}->{DIMS[1]}->{DIMS[2]}, 2 layers fwd+bwd, "
                               f"full graph per step, dropout 0",
                   "parallelism": parallelism, "launch": launch_mode},
        "bucket_ms": bucket_ms,
    }

    if events:
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
            kernels.append({"kernel": f"k_aggregate<{d // 4},{'true' if transposed else 'false'}>",
                            "d": d, "transposed": transposed, "launches_per_step": len(ts) // event_steps,
                            "avg_us": avg * 1e6, "bytes": nbytes, "gbs": nbytes / avg / 1e9,
                            "total_us_per_step": sum(ts) / event_steps * 1e6})
        dom = max(kernels, key=lambda k: k["total_us_per_step"])
        result["roofline"] = {"bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": dom["gbs"] / HBM_PEAK_GBS,
                              "traffic": pmc_traffic(dom["kernel"]) if world == 1 else None,
                              "kernel": dom["kernel"],
                              "avg_us": dom["avg_us"], "algorithmic_bytes_per_launch": dom["bytes"],
                              "event_bracket_overhead_us": event_overhead_us}
        result["gather_kernels"] = kernels

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

    if world > 1 and (backend == "nccl" or os.environ.get("RGCN_BENCH_TRY_GRAPH") == "1"):   # (the env: fallback rehearsal)
        # Everything above is measured and safe in `result`.  Now the captured-graph launch mode of
        # both N > 1 legs; the faster mode is the one reported.
        import threading
        safe_line = json.dumps(result)

        def bail_out():                                                    # pragma: no cover
            # the graph attempt hung (a collective that never completes): report what was measured
            if rank == 0:
                print(safe_line, flush=True)
            os._exit(0)

        watchdog = threading.Timer(float(os.environ.get("RGCN_BENCH_GRAPH_TIMEOUT", "120")), bail_out)
        watchdog.daemon = True
        watchdog.start()
        try:
            g_run, g_mode = pick_launch_mode(step)
            if g_mode != "eager":
                g_s = agree(timed(g_run, args.steps), dist.ReduceOp.MAX)
                if g_s < result["ms_per_step"] * 1e-3:
                    result.update(value=LAYERS * num_edges / g_s, ms_per_step=g_s * 1e3)
                    result["config"]["launch"] = g_mode
                    result["eager_ms_per_step"] = elapsed / args.steps * 1e3
            if "replica" in result:
                r_run, r_mode = pick_launch_mode(rep_step)
                if r_mode != "eager":
                    r_s = agree(timed(r_run, args.steps), dist.ReduceOp.MAX)
                    if r_s < result["replica"]["ms_per_step"] * 1e-3:
                        result["replica"].update(value=world * LAYERS * num_edges / r_s, ms_per_step=r_s * 1e3,
                                                 launch=r_mode)
        except Exception as exc:                                           # pragma: no cover
            print(f"bench: graph launch mode not measured ({exc!r}); reporting eager launches", file=sys.stderr)
            watchdog.cancel()
            # HIP is unusable in this process now: report the eager line and leave without teardown
            if rank == 0:
                print(safe_line, flush=True)
            sys.stderr.flush()
            os._exit(0)
        finally:
            watchdog.cancel()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(ei, et, n, r, args.cpu_seconds)
        result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:                                                  # pragma: no cover
            pass


if __name__ == "__main__":
    main()

"""Node-partitioned path (primekg_rgcn_linkprediction_amd/dist.py).

CPU tier: world_size-2 (and 3) gloo runs of the real partition / exchange / autograd logic
with an oracle-backed compute backend (the product backend is HIP only), checked against the
single-process oracle on the full graph.
GPU tier: the HIP backend on every rank's shard of a P-way partition, exchanges emulated by
concatenation in one process, checked bit-for-bit against the single-GPU path.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import need_gpu
from oracle import rgcn_oracle as O
from primekg_rgcn_linkprediction_amd import RGCNConv, ops, synth
from primekg_rgcn_linkprediction_amd import dist as rdist


class OracleBackend:
    """CPU stand-in for librgcn_hip.so built from plain torch ops (test infrastructure)."""

    def make_shard(self, key, other, etype, n_key, n_other, num_relations, edge_weight=None):
        return dict(key=key, other=other, et=etype, n_key=n_key, n_other=n_other, r=num_relations, w=edge_weight)

    def aggregate(self, s, x, out=None):
        assert x.size(0) == s["n_other"]
        dst = out
        seg = s["key"] * s["r"] + s["et"]
        rows = x[s["other"]]
        if s["w"] is not None:
            rows = rows * s["w"].view(-1, 1)
        out = x.new_zeros(s["n_key"] * s["r"], x.size(1)).index_add_(0, seg, rows)
        if s["w"] is None:
            cnt = torch.bincount(seg, minlength=s["n_key"] * s["r"]).clamp(min=1)
            out = out / cnt.view(-1, 1)
        res = out.view(s["n_key"], -1)
        return res if dst is None else dst.copy_(res)

    def transform_fwd(self, agg, x, weight, root, bias, relu=False, shard=None, table=None, out=None):
        res = agg @ weight.reshape(-1, weight.size(2))
        if root is not None:
            res = res + x @ root
        res = res + bias if bias is not None else res
        res = torch.relu(res) if relu else res
        return res if out is None else out.copy_(res)

    def transform_bwd_input(self, gagg, g, weight, root, relu_mask=None, shard=None, table=None, out=None):
        r, d_in, d_out = weight.shape
        gx = sum(gagg[:, k * d_out:(k + 1) * d_out] @ weight[k].t() for k in range(r))
        gx = gx + g @ root.t() if root is not None else gx
        gx = gx * (relu_mask > 0) if relu_mask is not None else gx
        return gx if out is None else out.copy_(gx)

    def transform_bwd_params(self, agg, x, g, num_relations, want_root, want_bias, shard=None):
        gw = (agg.t() @ g).view(num_relations, x.size(1), g.size(1))
        return gw, (x.t() @ g if want_root else None), (g.sum(0) if want_bias else None)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_problem(n, e, r, dims, seed):
    ei, et, _, _ = synth.uniform_graph(n, e, r, seed=seed)
    ei[1, : e // 10] = 3                                   # a heavy destination
    torch.manual_seed(seed)
    emb = torch.nn.init.xavier_uniform_(torch.empty(n, dims[0]))
    convs = [RGCNConv(dims[0], dims[1], r), RGCNConv(dims[1], dims[2], r)]
    for c in convs:
        c.bias.data.uniform_(-0.1, 0.1)
    cot = torch.randn(n, dims[2])
    return ei, et, emb, convs, cot


def _community_problem(world, n, e, r, dims, seed):
    """a graph with locality: `world` blocks of nodes, the first 60 % of each block connected inside the block only
    (interior rows under the block assignment), the rest also across blocks -> (problem, node -> rank)"""
    gen = torch.Generator().manual_seed(seed)
    per = n // world
    block = torch.randint(0, world, (e,), generator=gen)
    inner = int(per * 0.6)
    src = torch.randint(0, per, (e,), generator=gen) + block * per
    dst_local = torch.randint(0, per, (e,), generator=gen) + block * per
    dst_any = torch.randint(0, n, (e,), generator=gen)
    outer_src = (src % per) >= inner
    crossing = outer_src & (torch.rand(e, generator=gen) < 0.5) & ((dst_any % per) >= inner) & (dst_any < per * world)
    dst = torch.where(crossing, dst_any, dst_local)
    ei = torch.stack([src, dst])
    et = torch.randint(0, r, (e,), generator=gen)
    torch.manual_seed(seed)
    emb = torch.nn.init.xavier_uniform_(torch.empty(n, dims[0]))
    convs = [RGCNConv(dims[0], dims[1], r), RGCNConv(dims[1], dims[2], r)]
    for c in convs:
        c.bias.data.uniform_(-0.1, 0.1)
    cot = torch.randn(n, dims[2])
    assign = (torch.arange(n) // per).clamp(max=world - 1)
    return (ei, et, emb, convs, cot), assign


def _oracle_full(ei, et, emb, convs, cot):
    ps = [{k: v.detach().clone().requires_grad_(True) for k, v in c.named_parameters()} for c in convs]
    e = emb.clone().requires_grad_(True)
    out = O.encoder_ref(e, ps[0], ps[1], ei, et)
    (out * cot).sum().backward()
    return out.detach(), e.grad, ps


def _worker(rank, world, port, n, e, r, dims, seed, q, use_hip=False, scheme="pull", community=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assign = None
        if community:
            (ei, et, emb, convs, cot), assign = _community_problem(world, n, e, r, dims, seed)
        else:
            ei, et, emb, convs, cot = _make_problem(n, e, r, dims, seed)
        if use_hip:        # every rank drives the real kernels on the one GPU of the box
            dev = torch.device("cuda:0")
            torch.cuda.set_device(dev)
            enc = rdist.PartitionedEncoder(ei, et, n, r, emb, convs, dev, scheme=scheme, assignment=assign)
        else:
            dev = torch.device("cpu")
            enc = rdist.PartitionedEncoder(ei, et, n, r, emb, convs, dev, backend=OracleBackend(), scheme=scheme,
                                           assignment=assign)
        if community:      # the block assignment keeps whole neighbourhoods: the interior / boundary overlap is live
            summ0 = enc.exchange_summary()
            assert summ0["rows_interior"] >= 32
            assert summ0["interior_boundary_split"] == (scheme == "pull")
            if scheme == "push":
                assert 0 < summ0["push_rows_exchanged"] < n - summ0["rows_interior"]
        cot_own = enc.shard_rows(cot).to(dev)
        cot_own[enc.shard.num_own:] = 7.0            # junk in the padding slots must not reach any gradient
        out_own = enc.step(cot_own)
        if use_hip:        # the segments between exchanges become recorded passes (ops.Region) on their second and third
            first = [out_own.clone(), enc.emb.grad.clone()] + [p.grad.clone() for c in enc.convs for p in c.parameters()]
            for _ in range(4):                      # run: every later step - recorded, then replayed natively - same bits
                again = enc.step(cot_own)
                now = [again, enc.emb.grad] + [p.grad for c in enc.convs for p in c.parameters()]
                assert all(torch.equal(a, b) for a, b in zip(first, now))
            if scheme == "pull":
                plans = [v for g in (enc.shard.g_in, enc.shard.g_out, enc.shard.g_in_int, enc.shard.g_out_bnd) if g is not None
                         for v in g.__dict__.get("_regions", {}).values()]
                assert plans and all(isinstance(v, enc.backend.ops._Plan) for v in plans), [type(v) for v in plans]
        with torch.no_grad():                       # the per-layer nodes give the same rows
            assert torch.allclose(enc.forward_layers(), out_own, rtol=1e-6, atol=1e-6)
        out = enc.gather_output(out_own).cpu()
        gemb = enc.gather_output(enc.emb.grad).cpu()
        grads = {f"{i}.{k}": p.grad.cpu().numpy().copy() for i, c in enumerate(enc.convs)
                 for k, p in c.named_parameters()}
        summ = enc.exchange_summary()
        assert summ["halo_rows_backward"] <= summ["rows_remote"] and 0.0 <= summ["halo_fraction_backward"] <= 1.0
        balance = (enc.shard.num_in_edges, enc.shard.num_out_edges, enc.part.cap)
        # numpy payloads are pickled by value: no shared-memory handle outlives this process
        q.put((rank, out.numpy().copy(), gemb.numpy().copy(), grads, balance))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_partitioned(world, n, e, r, dims, seed, use_hip, scheme="pull", community=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, n, e, r, dims, seed, q, use_hip, scheme, community))
             for k in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results = [(rk, torch.from_numpy(o), torch.from_numpy(ge), {k: torch.from_numpy(v) for k, v in gr.items()}, b)
               for rk, o, ge, gr, b in results]
    return results


def _check_partitioned(results, world, n, e, r, dims, seed, community=False):
    if community:
        (ei, et, emb, convs, cot), _ = _community_problem(world, n, e, r, dims, seed)
    else:
        ei, et, emb, convs, cot = _make_problem(n, e, r, dims, seed)
    want_out, want_gemb, want_p = _oracle_full(ei, et, emb, convs, cot)
    total_in = 0
    for rank, out, gemb, grads, (n_in, n_out, cap) in results:
        torch.testing.assert_close(out, want_out, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(gemb, want_gemb, rtol=1e-4, atol=1e-5)
        for i in range(2):
            for k in ("weight", "root", "bias"):
                torch.testing.assert_close(grads[f"{i}.{k}"], want_p[i][k].grad, rtol=1e-4, atol=1e-4)
        assert cap == (n + world - 1) // world
        total_in += n_in
        assert n_in < 0.8 * e and n_out < 0.8 * e            # nobody holds (almost) everything
    assert total_in == e                                      # every edge has exactly one owner
    # every rank ends with identical (all-reduced) parameter gradients
    for k, v in results[0][3].items():
        for other in results[1:]:
            assert torch.equal(v, other[3][k])


@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("world,n,e", [(2, 101, 1500), (3, 64, 900)])
def test_partitioned_encoder_gloo(world, n, e, scheme):
    """both exchange schemes - "pull" (halo all-to-all-v of the rows a rank's edges read, owner computes) and
    "push" (the north star's form: partial sums from the source owner, reduce-scatter forward / all-gather
    backward) - against the single-process oracle on the full graph"""
    r, dims, seed = 3, (16, 32, 32), 5
    _check_partitioned(_run_partitioned(world, n, e, r, dims, seed, False, scheme), world, n, e, r, dims, seed)


@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_encoder_gloo_with_interior_rows(world, scheme):
    """a graph with locality under a block assignment (`assignment=`): most rows of a rank are INTERIOR, so "pull"
    runs their gather + transform behind the halo exchange and the boundary rows after it (two row ranges of one
    tensor), and "push" reduce-scatters only the rows another rank's sources reach - against the single-process
    oracle on the whole graph"""
    n, e, r, dims, seed = 150 * world, 2000 * world, 3, (16, 32, 32), 11
    res = _run_partitioned(world, n, e, r, dims, seed, False, scheme, community=True)
    _check_partitioned(res, world, n, e, r, dims, seed, community=True)


def test_interior_boundary_split_equals_the_unsplit_shard():
    """one rank's forward aggregate + transform and input gradient, interior rows / boundary rows as two row
    ranges (what runs around the halo exchange) against one pass over all rows - the oracle backend in float64:
    identical numbers; and the slots really are interior-first"""
    world, n, e, r = 3, 300, 4000, 3
    (ei, et, emb, convs, cot), assign = _community_problem(world, n, e, r, (16, 32, 32), 4)
    part = rdist.NodePartition.from_assignment(assign, ei, world)
    backend = OracleBackend()
    x, g = emb.double(), torch.randn(n, 32, dtype=torch.float64)
    w = torch.randn(r, 16, 32, dtype=torch.float64)
    root, bias = torch.randn(16, 32, dtype=torch.float64), torch.randn(32, dtype=torch.float64)
    cross = part.rank_of[ei[0]] != part.rank_of[ei[1]]
    touched = torch.zeros(n, dtype=torch.bool)
    touched[ei[0][cross]] = True
    touched[ei[1][cross]] = True
    for k in range(world):
        both = [rdist.RankShard(part, ei, et, r, k, torch.device("cpu"), backend, split=s) for s in (True, False)]
        sp, un = both
        assert sp.split and not un.split and sp.num_interior >= 32
        nodes = part.nodes_of(k)
        assert not touched[nodes[: sp.num_interior]].any() and touched[nodes[sp.num_interior:]].all()
        x_own, g_own = part.shard_rows(x, k), part.shard_rows(g, k)
        tbl = sp.halo_in.emulate(x_own, x)
        ki = sp.num_interior
        agg = torch.cat([backend.aggregate(sp.g_in_int, x_own), backend.aggregate(sp.g_in_bnd, tbl)])
        assert torch.equal(agg, backend.aggregate(un.g_in, tbl))
        out = torch.cat([backend.transform_fwd(agg[:ki], x_own[:ki], w, root, bias, True),
                         backend.transform_fwd(agg[ki:], x_own[ki:], w, root, bias, True)])
        assert torch.equal(out, backend.transform_fwd(agg, x_own, w, root, bias, True))
        gtbl = sp.halo_out.emulate(g_own, g)
        gagg = torch.cat([backend.aggregate(sp.g_out_int, g_own), backend.aggregate(sp.g_out_bnd, gtbl)])
        assert torch.equal(gagg, backend.aggregate(un.g_out, gtbl))


def _oracle_encoder(emb, ei, et, c1, c2):
    return O.encoder_ref(emb, dict(c1.named_parameters()), dict(c2.named_parameters()), ei, et)


def _replica_worker(rank, world, port, n, e, r, dims, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, et, emb, convs, _ = _make_problem(n, e, r, dims, seed)
        enc = rdist.ReplicatedEncoder(ei, et, n, r, emb, convs, torch.device("cpu"), encoder_fn=_oracle_encoder)
        cot = torch.randn(n, dims[2], generator=torch.Generator().manual_seed(100 + rank))   # this rank's batch
        enc.step(cot)
        q.put((rank, [p.grad.numpy().copy() for p in enc.params]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicated_encoder_gloo_averages_the_batches():
    """batch-replica mode, world 2: every rank ends with the mean over ranks of the gradients a
    single process computes for each rank's cotangent."""
    world, n, e, r, dims, seed = 2, 80, 1200, 3, (16, 32, 32), 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(k, world, port, n, e, r, dims, seed, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, et, emb, convs, _ = _make_problem(n, e, r, dims, seed)
    want = None
    for rank in range(world):
        cot = torch.randn(n, dims[2], generator=torch.Generator().manual_seed(100 + rank))
        _, gemb, ps = _oracle_full(ei, et, emb, convs, cot)
        grads = [gemb] + [ps[i][k].grad for i in range(2) for k, _ in convs[i].named_parameters()]
        want = grads if want is None else [a + b for a, b in zip(want, grads)]
    for rank in range(world):
        for g, w in zip(got[rank], want):
            torch.testing.assert_close(torch.from_numpy(g), w / world, rtol=1e-5, atol=1e-6)
    for a, b in zip(got[0], got[1]):
        assert (a == b).all()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["pull", "push"])
def test_partitioned_encoder_hip_two_ranks(scheme):
    """two processes, both on the box's one GPU, the product HIP backend in each; collectives
    over gloo (RCCL refuses two ranks on one device) - the full multi-process path minus RCCL."""
    need_gpu()
    world, n, e, r, dims, seed = 2, 3000, 60000, 3, (64, 128, 128), 8
    _check_partitioned(_run_partitioned(world, n, e, r, dims, seed, True, scheme), world, n, e, r, dims, seed)


def _replica_hip_worker(rank, world, port, n, e, r, dims, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ei, et, emb, convs, _ = _make_problem(n, e, r, dims, seed)
        enc = rdist.ReplicatedEncoder(ei, et, n, r, emb, convs, dev)          # product path: rgcn_encoder2 on the HIP library
        cot = torch.randn(n, dims[2], generator=torch.Generator().manual_seed(100 + rank)).to(dev)
        enc.step(cot)
        q.put((rank, [p.grad.cpu().numpy().copy() for p in enc.params]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_replicated_encoder_hip_two_ranks():
    """batch-replica mode with the HIP backend: two processes on the box's one GPU (gradient all-reduce
    host-staged over gloo), each its own cotangent; both end with the mean of the two oracle gradients."""
    need_gpu()
    world, n, e, r, dims, seed = 2, 2000, 40000, 3, (64, 128, 128), 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_hip_worker, args=(k, world, port, n, e, r, dims, seed, q)) for k in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ei, et, emb, convs, _ = _make_problem(n, e, r, dims, seed)
    want = None
    for rank in range(world):
        cot = torch.randn(n, dims[2], generator=torch.Generator().manual_seed(100 + rank))
        _, gemb, ps = _oracle_full(ei, et, emb, convs, cot)
        grads = [gemb] + [ps[i][k].grad for i in range(2) for k, _ in convs[i].named_parameters()]
        want = grads if want is None else [a + b for a, b in zip(want, grads)]
    for rank in range(world):
        for g, w in zip(got[rank], want):
            w = w / world
            assert ((torch.from_numpy(g) - w).abs().max() / (w.abs().max() + 1e-30)).item() <= 1e-4
    for a, b in zip(got[0], got[1]):
        assert (a == b).all()


def _rccl_worker(rank, world, port, n, e, r, dims, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        ei, et, emb, convs, cot = _make_problem(n, e, r, dims, seed)
        enc = rdist.PartitionedEncoder(ei, et, n, r, emb, convs, dev)
        out_own = enc.step(enc.shard_rows(cot).to(dev))
        out = enc.gather_output(out_own).cpu()
        gemb = enc.gather_output(enc.emb.grad).cpu()
        grads = {f"{i}.{k}": p.grad.cpu().numpy().copy() for i, c in enumerate(enc.convs) for k, p in c.named_parameters()}
        q.put((rank, out.numpy().copy(), gemb.numpy().copy(), grads,
               (enc.shard.num_in_edges, enc.shard.num_out_edges, enc.part.cap)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per device")
def test_partitioned_encoder_two_rccl_ranks():
    """PartitionedEncoder.step on 2 real RCCL ranks (one per GPU, halo all-to-all-v over xGMI) against the
    single-process oracle: runs wherever the box has two devices (the 1-GPU boxes of this pool skip it)."""
    world, n, e, r, dims, seed = 2, 3000, 60000, 3, (64, 128, 128), 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(k, world, port, n, e, r, dims, seed, q)) for k in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results = [(rk, torch.from_numpy(o), torch.from_numpy(ge), {k: torch.from_numpy(v) for k, v in gr.items()}, b)
               for rk, o, ge, gr, b in results]
    _check_partitioned(results, world, n, e, r, dims, seed)


def test_partition_is_balanced_and_consistent():
    ei, et, n, r = synth.primekg_like(num_edges=100000, seed=42)
    part = rdist.NodePartition(ei, n, 8)
    assert part.cap == (n + 7) // 8
    counts = torch.bincount(part.rank_of, minlength=8)
    assert counts.max() <= part.cap and counts.sum() == n
    assert torch.unique(part.pid).numel() == n and part.pid.max() < 8 * part.cap
    deg = torch.bincount(ei[1], minlength=n)
    per_rank = torch.zeros(8).index_add_(0, part.rank_of, deg.float())
    assert per_rank.max() / per_rank.mean() < 1.05            # edge-balanced despite the Zipf tail
    full = torch.arange(n * 2, dtype=torch.float32).view(n, 2)
    gathered = torch.cat([part.shard_rows(full, k) for k in range(8)])
    assert torch.equal(part.unshard_rows(gathered), full)


def test_clustered_partition_keeps_neighbourhoods_together():
    """``NodePartition.clustered``: on the PrimeKG-shaped graph most rows are interior only by
    accident under the degree-balanced deal; placing the light nodes WITH their neighbours nearly doubles the interior
    rows of every rank and cuts fewer edges, at the same row capacity and edge balance - so the interior-first
    overlap of the "pull" scheme has rows to work with.  A uniform random graph has none to find: "auto" keeps the deal."""
    ei, et, n, r = synth.primekg_like(seed=42)
    deg_in = torch.bincount(ei[1], minlength=n).float()
    for world in (2, 8):
        deal = rdist.NodePartition(ei, n, world)
        part = rdist.NodePartition.clustered(ei, n, world)
        assert part.cap == deal.cap and int(part.counts.max()) <= part.cap and int(part.counts.sum()) == n
        assert torch.unique(part.pid).numel() == n
        per_rank = torch.zeros(world).index_add_(0, part.rank_of, deg_in)
        assert per_rank.max() / per_rank.mean() <= 1.05
        assert int(part.num_interior.sum()) >= 1.4 * int(deal.num_interior.sum()), (world, part.num_interior, deal.num_interior)
        assert int(part.num_interior.min()) >= 1.4 * int(deal.num_interior.min())       # on EVERY rank, not one lucky one
        cut = lambda p: float((p.rank_of[ei[0]] != p.rank_of[ei[1]]).float().mean())    # noqa: E731
        assert cut(part) < cut(deal) - 0.05
        # interior means what RankShard relies on: no edge of an interior row, either direction, crosses ranks
        inner = part.interior
        assert not ((part.rank_of[ei[0]] != part.rank_of[ei[1]]) & (inner[ei[0]] | inner[ei[1]])).any()
    auto = rdist.NodePartition.build(ei, n, 8, "auto")
    assert torch.equal(auto.rank_of, rdist.NodePartition.clustered(ei, n, 8).rank_of)        # deterministic, and chosen
    ue, _, un, _ = synth.uniform_graph(20000, 400000, 4, seed=1)
    assert torch.equal(rdist.NodePartition.build(ue, un, 8, "auto").rank_of, rdist.NodePartition(ue, un, 8).rank_of)
    with pytest.raises(ValueError):
        rdist.NodePartition.build(ei, n, 8, "metis")


@pytest.mark.gpu
@pytest.mark.parametrize("world,method", [(2, "deal"), (4, "deal"), (8, "deal"), (8, "clustered")])
def test_hip_shards_match_single_gpu_bitwise(world, method, monkeypatch):
    """Each rank's bipartite structures on the real kernels; the all-gathers are emulated by
    concatenating the ranks' rows.  Activations and input grads must equal the 1-GPU path
    bit for bit; parameter grads (summed over ranks) to 1e-5.  (fp32 arithmetic: the split-precision
    transforms scale by the operand maximum of the rows a rank holds.)"""
    dev = need_gpu()
    monkeypatch.setattr(ops, "GEMM_PRECISION", "fp32")
    ei, et, n, r = synth.primekg_like(num_edges=60000, seed=7)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, 64, generator=gen)
    g = torch.randn(n, 128, generator=gen)
    conv = RGCNConv(64, 128, r).to(dev)
    conv.bias.data.uniform_(-0.1, 0.1)
    w, root, bias = conv.weight.detach(), conv.root.detach(), conv.bias.detach()
    # single GPU
    graph = ops.BucketedGraph(ei.to(dev), et.to(dev), n, r)
    agg1 = ops.aggregate(graph, x.to(dev))
    out1 = ops.transform_fwd(agg1, x.to(dev), w, root, bias)
    gx1 = ops.transform_bwd_input(ops.aggregate(graph, g.to(dev), transposed=True), g.to(dev), w, root)
    gw1, groot1, gbias1 = ops.transform_bwd_params(agg1, x.to(dev), g.to(dev), r)
    # P ranks
    backend = rdist.HipBackend()
    part = rdist.NodePartition.build(ei, n, world, method)      # (clustered: another owner for most rows, the same bits)
    xd, gd = x.to(dev), g.to(dev)
    outs, gxs, gw, groot, gbias = [], [], 0, 0, 0
    for k in range(world):
        shard = rdist.RankShard(part, ei, et, r, k, dev, backend)
        x_own, g_own = part.shard_rows(x, k).to(dev), part.shard_rows(g, k).to(dev)
        x_tbl = shard.halo_in.emulate(x_own, xd)          # [own rows | the rows the halo exchange would deliver]
        g_tbl = shard.halo_out.emulate(g_own, gd)
        assert shard.halo_in.num_halo <= n - shard.num_own and sum(shard.halo_in.recv_splits) == shard.halo_in.num_halo
        agg = backend.aggregate(shard.g_in, x_tbl)
        outs.append(backend.transform_fwd(agg, x_own, w, root, bias, False, shard.g_in, table=x_tbl))
        gxs.append(backend.transform_bwd_input(backend.aggregate(shard.g_out, g_tbl), g_own, w, root, None, shard.g_out,
                                               table=g_tbl))
        a, b, c = backend.transform_bwd_params(agg, x_own, g_own, r, True, True, shard.g_in)
        gw, groot, gbias = gw + a, groot + b, gbias + c
    assert torch.equal(part.unshard_rows(torch.cat(outs)), out1)
    assert torch.equal(part.unshard_rows(torch.cat(gxs)), gx1)
    for got, want in ((gw, gw1), (groot, groot1), (gbias, gbias1)):
        assert ((got - want).abs().max() / want.abs().max()).item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "split"])
def test_config_c4_node_partitioned_over_8_ranks_at_full_size(precision, monkeypatch):
    """BASELINE configs[3] as it is WRITTEN: 500,000 nodes / 20,000,000 edge columns / 16 relations, 64 -> 128,
    node-partitioned across 8 ranks.  One GPU builds each rank's shard in turn (partition, halo plans, both bucketed
    shard structures of its ~2.5 M edges - the previous one freed), runs the layer forward, the input gradient and the
    parameter gradients on it with the halo exchange emulated from the full tensors, and compares with the
    single-GPU run of the whole graph: >= 64 sampled output and grad_x rows per rank (bit for bit in fp32 arithmetic,
    where a rank's result cannot depend on which rows it holds; 1e-5 of the largest entry in split precision, where
    every rank scales by the maximum of ITS rows) and the parameter gradients summed over the ranks at 1e-5.  Edge
    balance <= 1.05; halo rows / bytes per rank and the peak memory go to gpurun_out/r04_c4_shards.txt."""
    import json
    dev = need_gpu()
    if torch.cuda.get_device_properties(dev).total_memory < 60 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    monkeypatch.setattr(ops, "GEMM_PRECISION", precision)
    world, d_in, d_out = 8, 64, 128
    ei, et, n, r = synth.uniform_graph(500_000, 20_000_000, 16, seed=42)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, d_in, generator=gen).to(dev)
    g = (torch.randn(n, d_out, generator=gen) * 1e-3).to(dev)
    torch.manual_seed(3)
    conv = RGCNConv(d_in, d_out, r).to(dev)
    conv.bias.data.uniform_(-0.1, 0.1)
    w, root, bias = conv.weight.detach(), conv.root.detach(), conv.bias.detach()
    eid, etd = ei.to(dev), et.to(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    # ---- the whole graph on the one GPU (separate kernels: the same entry points the shards take)
    graph = ops.BucketedGraph(eid, etd, n, r)
    amax_x = (ops.absmax(x),) * 2 if precision == "split" else None
    agg1 = ops.aggregate(graph, x)
    out1 = ops.transform_fwd(agg1, x, w, root, bias, amax=amax_x)
    gw1, groot1, gbias1 = ops.transform_bwd_params(agg1, x, g, r, graph=graph)
    del agg1
    amax_g = (ops.absmax(g),) * 2 if precision == "split" else None
    gagg1 = ops.aggregate(graph, g, transposed=True)
    gx1 = ops.transform_bwd_input(gagg1, g, w, root, amax=amax_g, amax_mul=graph.weight_bound(True))
    del gagg1
    graph.destroy()
    torch.cuda.synchronize()
    peak_single = torch.cuda.max_memory_allocated(dev)
    # ---- 8 ranks, one after the other
    backend = rdist.HipBackend()
    part = rdist.NodePartition(ei, n, world)
    deg = torch.bincount(ei[1], minlength=n).float()
    per_rank_edges = torch.zeros(world).index_add_(0, part.rank_of, deg)
    balance = float(per_rank_edges.max() / per_rank_edges.mean())
    assert balance <= 1.05, balance
    report = {"config": "C4 500000 / 20000000 / 16, 64 -> 128, P = 8 (degree-balanced deal), one GPU, shards in turn",
              "precision": precision, "edge_balance_max_over_mean": balance, "cap_rows": part.cap, "ranks": []}
    gw, groot, gbias = 0, 0, 0
    pick = torch.Generator().manual_seed(11)
    worst_out = worst_gx = 0.0
    for k in range(world):
        torch.cuda.reset_peak_memory_stats(dev)
        shard = rdist.RankShard(part, eid, etd, r, k, dev, backend)
        nodes = part.nodes_of(k).to(dev)
        x_own, g_own = x.new_zeros(part.cap, d_in), g.new_zeros(part.cap, d_out)
        x_own[: nodes.numel()], g_own[: nodes.numel()] = x[nodes], g[nodes]
        x_tbl = shard.halo_in.emulate(x_own, x)          # [own rows | the rows the halo exchange would deliver]
        g_tbl = shard.halo_out.emulate(g_own, g)
        agg = backend.aggregate(shard.g_in, x_tbl)
        out = backend.transform_fwd(agg, x_own, w, root, bias, False, shard.g_in, table=x_tbl)
        gagg = backend.aggregate(shard.g_out, g_tbl)
        gx = backend.transform_bwd_input(gagg, g_own, w, root, None, shard.g_out, table=g_tbl)
        del gagg
        a, b, c = backend.transform_bwd_params(agg, x_own, g_own, r, True, True, shard.g_in)
        gw, groot, gbias = gw + a, groot + b, gbias + c
        rows = torch.randperm(nodes.numel(), generator=pick)[:96].to(dev)          # >= 64 sampled rows of this rank
        for got, want, tag in ((out, out1, "out"), (gx, gx1, "gx")):
            gr, wr = got[rows], want[nodes[rows]]
            if precision == "fp32":
                assert torch.equal(gr, wr), (k, tag)
            else:
                err = float((gr - wr).abs().max() / want.abs().max())
                assert err <= 1e-5, (k, tag, err)
                if tag == "out":
                    worst_out = max(worst_out, err)
                else:
                    worst_gx = max(worst_gx, err)
        torch.cuda.synchronize()
        report["ranks"].append({
            "rank": k, "own_rows": shard.num_own, "interior_rows": shard.num_interior,
            "in_edges": shard.num_in_edges, "out_edges": shard.num_out_edges,
            "halo_rows_forward": shard.halo_in.num_halo, "halo_rows_backward": shard.halo_out.num_halo,
            "halo_bytes_forward_layer1": shard.halo_in.num_halo * d_in * 4,
            "halo_bytes_backward_layer1": shard.halo_out.num_halo * d_out * 4,
            "send_rows_forward": shard.halo_in.num_send,
            "peak_device_bytes": int(torch.cuda.max_memory_allocated(dev))})
        del shard, agg, out, gx, x_tbl, g_tbl, x_own, g_own
        ops.clear_graph_cache()
    for got, want in ((gw, gw1), (groot, groot1), (gbias, gbias1)):
        assert ((got - want).abs().max() / want.abs().max()).item() < 1e-5
    report["single_gpu_peak_device_bytes"] = int(peak_single)
    report["worst_sampled_row_error_vs_single_gpu"] = {"out": worst_out, "grad_x": worst_gx}
    try:
        root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root_dir, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root_dir, "gpurun_out", f"r04_c4_shards_{precision}.txt"), "w") as f:
            f.write(json.dumps(report, indent=1) + "\n")
    except OSError:
        pass


@pytest.mark.gpu
def test_hip_interior_boundary_split_is_bitwise_the_unsplit_shard(monkeypatch):
    """the interior / boundary halves on the real kernels (fp32 arithmetic, where the scale of an operand does not
    enter): aggregate, layer output and input gradient of every rank equal the one-pass shard bit for bit - the
    halves only change WHEN rows are computed (interior rows behind the halo exchange), not how"""
    dev = need_gpu()
    monkeypatch.setattr(ops, "GEMM_PRECISION", "fp32")
    world, n, e, r = 4, 4000, 90000, 3
    (ei, et, emb, _, _), assign = _community_problem(world, n, e, r, (64, 128, 128), 6)
    part = rdist.NodePartition.from_assignment(assign, ei, world)
    backend = rdist.HipBackend()
    gen = torch.Generator().manual_seed(2)
    x, g = emb, torch.randn(n, 128, generator=gen)
    conv = RGCNConv(64, 128, r).to(dev)
    conv.bias.data.uniform_(-0.1, 0.1)
    w, root, bias = conv.weight.detach(), conv.root.detach(), conv.bias.detach()
    mask = torch.randn(n, 64, generator=gen)
    xd, gd = x.to(dev), g.to(dev)
    for k in range(world):
        sp = rdist.RankShard(part, ei, et, r, k, dev, backend, split=True)
        un = rdist.RankShard(part, ei, et, r, k, dev, backend, split=False)
        assert sp.split and sp.num_interior >= 32
        ki = sp.num_interior
        x_own, g_own, m_own = (part.shard_rows(t, k).to(dev) for t in (x, g, mask))
        tbl, gtbl = sp.halo_in.emulate(x_own, xd), sp.halo_out.emulate(g_own, gd)
        agg = torch.empty(part.cap, r * 64, device=dev)
        out = torch.empty(part.cap, 128, device=dev)
        backend.aggregate(sp.g_in_int, x_own, out=agg[:ki])
        backend.transform_fwd(agg[:ki], x_own[:ki], w, root, bias, True, sp.g_in_int, table=x_own, out=out[:ki])
        backend.aggregate(sp.g_in_bnd, tbl, out=agg[ki:])
        backend.transform_fwd(agg[ki:], x_own[ki:], w, root, bias, True, sp.g_in_bnd, table=tbl, out=out[ki:])
        agg_u = backend.aggregate(un.g_in, tbl)
        assert torch.equal(agg, agg_u)
        assert torch.equal(out, backend.transform_fwd(agg_u, x_own, w, root, bias, True, un.g_in, table=tbl))
        gagg = torch.empty(part.cap, r * 128, device=dev)
        gx = torch.empty(part.cap, 64, device=dev)
        backend.aggregate(sp.g_out_int, g_own, out=gagg[:ki])
        backend.transform_bwd_input(gagg[:ki], g_own[:ki], w, root, m_own[:ki], sp.g_out_int, table=g_own, out=gx[:ki])
        backend.aggregate(sp.g_out_bnd, gtbl, out=gagg[ki:])
        backend.transform_bwd_input(gagg[ki:], g_own[ki:], w, root, m_own[ki:], sp.g_out_bnd, table=gtbl, out=gx[ki:])
        gagg_u = backend.aggregate(un.g_out, gtbl)
        assert torch.equal(gagg, gagg_u)
        assert torch.equal(gx, backend.transform_bwd_input(gagg_u, g_own, w, root, m_own, un.g_out, table=gtbl))


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["pull", "push"])
def test_partitioned_encoder_hip_two_ranks_with_interior_rows(scheme):
    """two processes on the box's one GPU, the HIP backend, a block assignment with interior rows: the overlapped
    "pull" halves and the boundary-only "push" exchange end to end against the oracle"""
    need_gpu()
    world, n, e, r, dims, seed = 2, 3000, 60000, 3, (64, 128, 128), 12
    res = _run_partitioned(world, n, e, r, dims, seed, True, scheme, community=True)
    _check_partitioned(res, world, n, e, r, dims, seed, community=True)


def test_partition_bulk_tail_keeps_capacity_and_balance():
    """nodes past the exact head are dealt in bulk (vectorised): every node gets one slot, nobody exceeds the
    capacity, the loads stay balanced; below the head size the deal is the exact greedy one"""
    ei, et, n, r = synth.primekg_like(num_edges=100000, seed=42)
    exact = rdist.NodePartition(ei, n, 8)
    for head in (0, 500, 5000):
        part = rdist.NodePartition(ei, n, 8, exact_head=head)
        assert torch.unique(part.pid).numel() == n and int(part.slot_of.max()) < part.cap
        assert int(part.counts.max()) <= part.cap and int(part.counts.sum()) == n
        deg = torch.bincount(ei[1], minlength=n).float()
        per_rank = torch.zeros(8).index_add_(0, part.rank_of, deg)
        assert per_rank.max() / per_rank.mean() < (1.05 if head >= 500 else 1.6)
    same = rdist.NodePartition(ei, n, 8, exact_head=n)
    assert torch.equal(same.rank_of, exact.rank_of) and torch.equal(same.slot_of, exact.slot_of)
    # halo plans of all ranks fit together: what q sends to p is what p expects from q
    class _Rec:
        def make_shard(self, *a, **k):
            return a
    shards = [rdist.RankShard(exact, ei, et, r, k, torch.device("cpu"), _Rec()) for k in range(8)]
    for p in range(8):
        for q_ in range(8):
            assert shards[p].halo_in.recv_splits[q_] == shards[q_].halo_in.send_splits[p]
            assert shards[p].halo_out.recv_splits[q_] == shards[q_].halo_out.send_splits[p]
        assert 0 < shards[p].halo_fraction_in < 1                      # a Zipf graph: a real halo, not the whole table


@pytest.mark.gpu
def test_partitioned_encoder_single_rank_nccl():
    """world_size 1 over RCCL: the full PartitionedEncoder code path on the HIP backend."""
    dev = need_gpu()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, e, r, dims = 500, 8000, 3, (64, 128, 128)
        ei, et, emb, convs, cot = _make_problem(n, e, r, dims, 3)
        want_out, want_gemb, want_p = _oracle_full(ei, et, emb, convs, cot)
        enc = rdist.PartitionedEncoder(ei, et, n, r, emb, convs, dev)
        out = enc.gather_output(enc.step(enc.shard_rows(cot).to(dev))).cpu()
        assert (out - want_out).abs().max() <= 1e-5
        gemb = enc.gather_output(enc.emb.grad).cpu()
        assert ((gemb - want_gemb).abs().max() / want_gemb.abs().max()) < 1e-4
        for i, c in enumerate(enc.convs):
            for k, p in c.named_parameters():
                ref = want_p[i][k].grad
                assert ((p.grad.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4
    finally:
        dist.destroy_process_group()

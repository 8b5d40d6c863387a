"""Node-partitioned execution of the R-GCN encoder: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI), one exchange per layer and
direction.  The reference has no distributed code at all; this is the
multi-GPU form BASELINE.json's north star asks for, built for MI355X's point-to-point xGMI.

Partition: nodes are dealt to the P ranks greedily in descending degree (least-loaded rank first,
at most cap = ceil(N/P) row slots each): near-equal shares of edge endpoints (PrimeKG ids are
type-sorted and genes carry ~88 % of the endpoints, so contiguous ranges would be badly skewed,
and its hubs defeat a plain round-robin deal).  Built once on rank 0 and broadcast.

Two exchange schemes, switchable (``scheme=`` / ``RGCN_DIST_SCHEME``), same results:

``"pull"`` (default; SURVEY.md section 8e "alternative worth measuring") - owner computes:
  rank p holds the in-edges of its rows (forward structure, mean mode) and the out-edges of its
  rows (transposed structure, weights 1/cnt[dst, rel] from the GLOBAL counts).
  forward of a layer : HALO exchange of x - every rank receives exactly the rows its edges read
                       from every peer (one all-to-all-v; the lists are fixed with the graph) ->
                       gather+mean over own (dst, rel) segments -> transform of own rows only;
  backward           : halo exchange of g -> weighted gather over own (src, rel) segments ->
                       input-grad transform of own rows.
  No row of the output is ever a cross-rank partial sum, so activations and input gradients are
  bit-identical to the single-GPU run in fp32 arithmetic (same per-segment summation order, same
  k-ordered MFMA chains); BOTH halves of the layer are sharded.
``"push"`` (the north star's form) - source owner computes partial sums:
  rank q holds the edges whose SOURCE it owns, bucketed by (dst, rel) over all N rows, weights
  1/cnt; forward: partial aggregates of all rows from own x -> transform (linear, so partial
  outputs add) -> REDUCE-SCATTER of the [P*cap, d_out] partial outputs -> the owner adds
  x root + bias; backward: ALL-GATHER of g (the other half of the all-reduce) -> the same
  transposed gather as "pull".  Shards the gather only: every rank transforms all N rows.

Parameter gradients are partial sums over a rank's rows -> ONE flat all-reduce per layer
(<= 0.4 MB), overlapped with the gather that follows.

Overlap ("pull"): a rank's slots hold its INTERIOR rows first - rows none of whose edges (either
direction) crosses ranks - then its boundary rows.  Per layer and direction the halo exchange is issued
first; gather + transform of the interior rows (they read own rows only) and, in backward, the
parameter-gradient GEMM run while it is in flight; the boundary rows follow the wait.  Both halves
write row ranges of ONE aggregate / output tensor, in the same per-segment order as the unsplit form
(fp32 arithmetic: the same bits).  How many rows are interior is the partition's doing: the
degree-balanced deal below scatters neighbours over all ranks (C2, P = 8: a handful of low-degree
rows), a locality-aware assignment (``NodePartition.from_assignment``) keeps whole neighbourhoods.
"push" reduce-scatters only the rows that receive a message from another rank's sources.

xGMI is point to point (7 links per GPU): an all-to-all / all-gather / reduce-scatter of slabs
drives all seven links at once.

The compute backend is injectable so that the N > 1 logic is covered by world_size-2 gloo
tests on CPU (tests/ supply an oracle-backed backend); the product backend is the HIP
library and there is no CPU fallback here.
"""
from __future__ import annotations

import heapq
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor

SCHEMES = ("pull", "push")


class HipBackend:
    """The product compute path: every call lands in librgcn_hip.so.  The per-layer segments between two exchanges
    (gather + transform of a row range; gather + input-gradient transform) are ``ops.Region`` passes: recorded on
    their second and third run over a shard structure, then issued by one native call each - at N > 1 a rank's
    kernels shrink with 1 / N while the host cost of issuing them does not."""

    def __init__(self):
        from . import ops
        self.ops = ops
        self._fwd = ops.Region("dist.layer_fwd", self._fwd_pass)
        self._fwd_rows = ops.Region("dist.layer_fwd_rows", self._fwd_rows_pass)
        self._bwd = ops.Region("dist.input_grad", self._bwd_pass)
        self._bwd_rows = ops.Region("dist.input_grad_rows", self._bwd_rows_pass)

    def make_shard(self, key, other, etype, n_key, n_other, num_relations, edge_weight=None):
        return self.ops.BucketedGraph.from_shard(key, other, etype, n_key, n_other, num_relations,
                                                 edge_weight)

    def aggregate(self, shard, x, out=None):
        return self.ops.aggregate(shard, x, out=out)

    def _amax(self, table, shard):
        """operand scales of the split-precision transforms: the aggregate is bounded by
        (largest per-segment weight sum) * max |table it was gathered from|; the rank's own rows are
        part of that table, so its maximum serves both operands - one launch, no collective"""
        if table is None or self.ops.GEMM_PRECISION != "split":
            return None, 1.0
        t = self.ops.absmax(table)
        return (t, t), (shard.weight_bound(False) if shard is not None else 1.0)

    # `shard` = the structure the aggregate operand was built over (its relation-occupancy mask
    # lets the kernels skip all-zero tiles), `table` = the rows it was gathered from; backends
    # without these notions ignore them
    def transform_fwd(self, agg, x, weight, root, bias, relu=False, shard=None, table=None, out=None):
        amax, mul = self._amax(table, shard)
        return self.ops.transform_fwd(agg, x, weight, root, bias, relu, shard, amax=amax, amax_mul=mul, out=out)

    def transform_bwd_input(self, gagg, g, weight, root, relu_mask=None, shard=None, table=None, out=None):
        amax, mul = self._amax(table, shard)
        gx = self.ops.transform_bwd_input(gagg, g, weight, root, relu_mask, shard, amax=amax, amax_mul=mul)
        if out is None:
            return gx
        self.ops.guard_torch_op("row-range copy of an input gradient")
        out.copy_(gx)                     # (the input-gradient entry points allocate their result: one row-range copy)
        return out

    def transform_bwd_params(self, agg, x, g, num_relations, want_root, want_bias, shard=None):
        return self.ops.transform_bwd_params(agg, x, g, num_relations, want_root, want_bias, shard)

    # ---- whole segments (what _layer_fwd / _input_grad call when the backend has them) ------------------------
    def _fwd_pass(self, tbl, x, weight, root, bias, *, shard, relu):
        agg = self.ops.aggregate(shard, tbl)
        return self.transform_fwd(agg, x, weight, root, bias, relu, shard, table=tbl), agg

    def _fwd_rows_pass(self, tbl, x, weight, root, bias, agg, out, *, shard, relu, lo, hi):
        """rows [lo, hi) of `agg` / `out` (allocated by the caller when lo == 0 is not the first call)"""
        self.ops.aggregate(shard, tbl, out=agg[lo:hi])
        self.transform_fwd(agg[lo:hi], x[lo:hi], weight, root, bias, relu, shard, table=tbl, out=out[lo:hi])
        return ()

    def _bwd_pass(self, tbl, g, weight, root, mask, *, shard):
        gagg = self.ops.aggregate(shard, tbl)
        return (self.transform_bwd_input(gagg, g, weight, root, mask, shard, table=tbl),)

    def _bwd_rows_pass(self, tbl, g, weight, root, mask, gagg, *, shard, lo, hi):
        self.ops.aggregate(shard, tbl, out=gagg[lo:hi])
        m = mask[lo:hi] if mask is not None else None
        return (self.transform_bwd_input(gagg[lo:hi], g[lo:hi], weight, root, m, shard, table=tbl),)

    @staticmethod
    def _key(*tensors, extra=()):
        return tuple(tuple(t.shape) if t is not None else None for t in tensors) + tuple(extra)

    def layer_fwd(self, shard, tbl, x, weight, root, bias, relu):
        out, agg = self._fwd.run(shard, self._key(tbl, x, weight, root, bias, extra=(relu,)), (tbl, x, weight, root, bias),
                                 dict(shard=shard, relu=relu), want={0, 1})
        return out, agg

    def layer_fwd_rows(self, shard, tbl, x, weight, root, bias, relu, agg, out, lo, hi):
        self._fwd_rows.run(shard, self._key(tbl, x, weight, root, bias, agg, out, extra=(relu, lo, hi)),
                           (tbl, x, weight, root, bias, agg, out), dict(shard=shard, relu=relu, lo=lo, hi=hi))

    def input_grad(self, shard, tbl, g, weight, root, mask):
        return self._bwd.run(shard, self._key(tbl, g, weight, root, mask), (tbl, g, weight, root, mask), dict(shard=shard),
                             want={0})[0]

    def input_grad_rows(self, shard, tbl, g, weight, root, mask, gagg, lo, hi):
        return self._bwd_rows.run(shard, self._key(tbl, g, weight, root, mask, gagg, extra=(lo, hi)),
                                  (tbl, g, weight, root, mask, gagg), dict(shard=shard, lo=lo, hi=hi), want={0})[0]


# ------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------
class NodePartition:
    """Deterministic assignment node -> (rank, slot).  ``NodePartition(...)`` computes it locally;
    ``NodePartition.shared(...)`` computes it on rank 0 and broadcasts (every rank of a job then
    holds the identical tensors without each running the serial part)."""

    EXACT_HEAD = 65536      # nodes (heaviest first) dealt by the exact heap; the light tail is dealt in bulk

    def __init__(self, edge_index: Tensor, num_nodes: int, world: int, exact_head: Optional[int] = None):
        ei = edge_index.cpu()
        deg = torch.bincount(ei[0], minlength=num_nodes) + torch.bincount(ei[1], minlength=num_nodes)
        self.world, self.num_nodes = world, num_nodes
        self.cap = (num_nodes + world - 1) // world
        rank_of, slot_of = self._deal(deg, world, self.cap, self.EXACT_HEAD if exact_head is None else exact_head)
        self._finish(*self._interior_first(rank_of, slot_of, ei, self.cap))

    @classmethod
    def from_assignment(cls, rank_of: Tensor, edge_index: Tensor, world: int) -> "NodePartition":
        """a partition from a given node -> rank assignment (a locality-aware partitioner's output): slots in node
        order, interior rows first; every rank at most ceil(N / world) rows"""
        part = cls.__new__(cls)
        rank_of = rank_of.to(torch.int64).cpu()
        n = rank_of.numel()
        part.world, part.num_nodes, part.cap = world, n, (n + world - 1) // world
        counts = torch.bincount(rank_of, minlength=world)
        if int(counts.max()) > part.cap:
            raise ValueError(f"a rank holds {int(counts.max())} rows, more than ceil(N / world) = {part.cap}")
        order = torch.argsort(rank_of, stable=True)
        start = torch.cumsum(counts, 0) - counts
        slot_of = torch.empty(n, dtype=torch.int64)
        slot_of[order] = torch.arange(n) - start[rank_of[order]]
        part._finish(*cls._interior_first(rank_of, slot_of, edge_index.cpu(), part.cap))
        return part

    CLUSTER_MAX_NODES = 200_000    # `clustered` walks the nodes in Python: PrimeKG-sized graphs, not C4

    @classmethod
    def clustered(cls, edge_index: Tensor, num_nodes: int, world: int, balance: float = 1.04,
                  heavy_share: float = 0.7) -> "NodePartition":
        """A locality-aware assignment for degree-skewed graphs (PrimeKG: a few hub genes, a long tail of nodes with
        one or two neighbours), so that the interior-first overlap of the "pull" scheme has rows to work with: under
        the degree-balanced deal a row is interior only if ALL its neighbours happened to land on its rank - a handful
        of rows at P = 8 - although most nodes have so few neighbours that they could simply sit WITH them.

        1. the heaviest nodes - those that together hold `heavy_share` of the edge endpoints - are dealt by the exact
           longest-processing-time heap of `_deal` (they are what balances the edge load);
        2. every other node, heaviest first (its heavier neighbours are placed by then), goes to the rank that already
           holds most of its neighbours, among the ranks with a free row slot whose edge load stays under `balance` x
           the mean; ties and neighbourless nodes go to the least-loaded such rank.

        Same capacity (ceil(N / world) rows per rank) and slot conventions as the deal, so everything downstream -
        halo plans, shards, results - is unchanged: only WHICH rows a rank owns differs (results in fp32 arithmetic
        are bit-identical to any other assignment's: no output row is a cross-rank sum)."""
        import numpy as np
        if num_nodes > cls.CLUSTER_MAX_NODES:
            raise ValueError(f"clustered() walks the nodes one by one: {num_nodes} > {cls.CLUSTER_MAX_NODES}; use the deal")
        ei = edge_index.cpu()
        src, dst = ei[0].numpy(), ei[1].numpy()
        n = num_nodes
        deg = np.bincount(src, minlength=n) + np.bincount(dst, minlength=n)
        cap = (n + world - 1) // world
        # undirected neighbour lists (CSR), duplicates kept: a neighbour reached by more edges pulls harder
        a = np.concatenate([src, dst])
        b = np.concatenate([dst, src])
        order = np.argsort(a, kind="stable")
        nbr = b[order]
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(a, minlength=n), out=ptr[1:])
        by_deg = np.argsort(-deg, kind="stable")
        total = float(deg.sum())
        cum = np.cumsum(deg[by_deg])
        heavy = int(np.searchsorted(cum, heavy_share * total)) + 1 if total > 0 else 0
        heavy = min(n, max(heavy, world))
        rank_of = np.full(n, -1, dtype=np.int64)
        load = np.zeros(world, dtype=np.float64)
        fill = np.zeros(world, dtype=np.int64)
        heap = [(0.0, k) for k in range(world)]
        for v in by_deg[:heavy].tolist():                                # step 1: exact LPT over the heavy head
            ld, k = heapq.heappop(heap)
            rank_of[v] = k
            fill[k] += 1
            load[k] = ld + deg[v]
            if fill[k] < cap:
                heapq.heappush(heap, (load[k], k))
        limit = balance * total / world
        for v in by_deg[heavy:].tolist():                                # step 2: join the neighbours
            open_ = fill < cap
            ok = open_ & (load + deg[v] <= limit)
            if not ok.any():
                ok = open_
            votes = np.zeros(world, dtype=np.int64)
            rk = rank_of[nbr[ptr[v]:ptr[v + 1]]]
            rk = rk[rk >= 0]
            if rk.size:
                votes = np.bincount(rk, minlength=world)
            score = np.where(ok, votes.astype(np.float64) - 1e-9 * load, -np.inf)   # most neighbours; then least loaded
            k = int(np.argmax(score))
            rank_of[v] = k
            fill[k] += 1
            load[k] += deg[v]
        return cls.from_assignment(torch.from_numpy(rank_of), ei, world)

    @staticmethod
    def _interior_first(rank_of: Tensor, slot_of: Tensor, ei: Tensor, cap: int):
        """re-number each rank's slots so that its interior rows (no edge of theirs, in either direction, crosses
        ranks) come first, each class in the old slot order -> (rank_of, slot_of, interior flag)"""
        n = rank_of.numel()
        cross = rank_of[ei[0]] != rank_of[ei[1]]
        boundary = torch.zeros(n, dtype=torch.bool)
        boundary[ei[0][cross]] = True
        boundary[ei[1][cross]] = True
        order = torch.argsort(rank_of * (2 * cap) + boundary.long() * cap + slot_of, stable=True)
        counts = torch.bincount(rank_of, minlength=int(rank_of.max()) + 1 if n else 1)
        start = torch.cumsum(counts, 0) - counts
        new_slot = torch.empty(n, dtype=torch.int64)
        new_slot[order] = torch.arange(n) - start[rank_of[order]]
        return rank_of, new_slot, ~boundary

    @classmethod
    def shared(cls, edge_index: Tensor, num_nodes: int, group=None, device=None, method: str = "deal") -> "NodePartition":
        """``method``: "deal" (degree-balanced), "clustered" (locality-aware, PrimeKG-sized graphs) or "auto"
        (clustered up to CLUSTER_MAX_NODES nodes on a skewed graph, else the deal)"""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if rank == 0:
            part = cls.build(edge_index, num_nodes, world, method)
            payload = torch.stack([part.rank_of, part.slot_of, part.interior.long()])
        else:
            part = cls.__new__(cls)
            part.world, part.num_nodes, part.cap = world, num_nodes, (num_nodes + world - 1) // world
            payload = torch.empty(3, num_nodes, dtype=torch.int64)
        if world > 1:
            on_dev = dist.get_backend(group) == "nccl"
            buf = payload.to(device) if on_dev else payload
            dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            payload = buf.cpu()
        if rank != 0:
            part._finish(payload[0].clone(), payload[1].clone(), payload[2].bool())
        return part

    @classmethod
    def build(cls, edge_index: Tensor, num_nodes: int, world: int, method: str = "deal") -> "NodePartition":
        if method not in ("deal", "clustered", "auto"):
            raise ValueError(f"partition method must be 'deal', 'clustered' or 'auto', got {method!r}")
        if method == "auto":
            ei = edge_index.cpu()
            deg = torch.bincount(ei[0], minlength=num_nodes) + torch.bincount(ei[1], minlength=num_nodes)
            # clustering pays where most nodes have a handful of neighbours (they can sit with them); a uniform
            # random graph (C4: every node ~80 neighbours on all ranks) has no interior rows under any assignment
            light = float((deg <= 8).float().mean()) if num_nodes else 0.0
            method = "clustered" if (world > 1 and num_nodes <= cls.CLUSTER_MAX_NODES and light >= 0.15) else "deal"
        part = cls.clustered(edge_index, num_nodes, world) if method == "clustered" else cls(edge_index, num_nodes, world)
        part.method = method
        return part

    def _finish(self, rank_of: Tensor, slot_of: Tensor, interior: Tensor) -> None:
        self.rank_of, self.slot_of, self.interior = rank_of, slot_of, interior
        self.pid = rank_of * self.cap + slot_of                         # row in the gathered layout
        self.counts = torch.bincount(rank_of, minlength=self.world)
        # rows [0, num_interior[k]) of rank k are interior, [num_interior[k], counts[k]) boundary
        self.num_interior = torch.bincount(rank_of[interior], minlength=self.world)
        self._dev = {}

    def on(self, device) -> "_PartitionTensors":
        """rank_of / slot_of / pid on ``device`` (cached): the shard plans are index arithmetic over the whole edge
        list - 20M columns at C4 - and run where the edges are"""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = _PartitionTensors(self.rank_of.to(device), self.slot_of.to(device), self.pid.to(device))
        return self._dev[key]

    @staticmethod
    def _deal(deg: Tensor, world: int, cap: int, exact_head: int):
        """Longest-processing-time greedy under a capacity of `cap` rows per rank: each node, in
        descending degree, goes to the least-loaded rank that still has a free slot (ties -> lowest
        rank).  A Zipf tail (top node ~4 % of all endpoints) defeats a plain round-robin deal; this keeps
        max/mean edge load within a few percent.  The heap loop is inherently serial, so only the
        `exact_head` heaviest nodes take it (all of PrimeKG); the light tail - loads are within one
        head-sized degree of each other by then - is dealt in bulk: rounds over the ranks in ascending
        load order, each rank taking one node per round until its slots are full."""
        n = deg.numel()
        order = torch.argsort(deg, descending=True, stable=True)        # heavy nodes first
        head = min(n, max(0, exact_head))
        heap = [(0, k) for k in range(world)]
        fill = [0] * world
        load = [0] * world
        rank_l, slot_l = [0] * head, [0] * head
        for i, d in enumerate(deg[order[:head]].tolist()):
            ld, k = heapq.heappop(heap)
            rank_l[i], slot_l[i] = k, fill[k]
            fill[k] += 1
            load[k] = ld + d
            if fill[k] < cap:
                heapq.heappush(heap, (ld + d, k))
        rank_of = torch.empty(n, dtype=torch.int64)
        slot_of = torch.empty(n, dtype=torch.int64)
        rank_of[order[:head]] = torch.tensor(rank_l, dtype=torch.int64)
        slot_of[order[:head]] = torch.tensor(slot_l, dtype=torch.int64)
        tail = n - head
        if tail > 0:
            fill_t = torch.tensor(fill, dtype=torch.int64)
            # final row counts: as even as the capacity allows (the first n - (cap-1)*world ranks get cap)
            base = n // world
            final = torch.full((world,), base, dtype=torch.int64)
            final[: n - base * world] += 1
            quota = (final - fill_t).clamp(min=0)
            short = tail - int(quota.sum())                              # head dealt unevenly: hand the rest to
            k = 0                                                        # the ranks with free slots, lightest first
            by_load = sorted(range(world), key=lambda r: (load[r], r))
            while short != 0:
                r = by_load[k % world]
                if short > 0 and fill_t[r] + quota[r] < cap:
                    quota[r] += 1
                    short -= 1
                elif short < 0 and quota[r] > 0:
                    quota[r] -= 1
                    short += 1
                k += 1
            pos_in_order = torch.empty(world, dtype=torch.int64)
            pos_in_order[torch.tensor(by_load)] = torch.arange(world)
            rk = torch.repeat_interleave(torch.arange(world), quota)                    # owner of each tail slot
            start = torch.cumsum(quota, 0) - quota
            rnd = torch.arange(tail) - torch.repeat_interleave(start, quota)           # its round
            seq = torch.argsort(rnd * world + pos_in_order[rk], stable=True)            # rounds, ranks by ascending load
            rank_of[order[head:]] = rk[seq]
            slot_of[order[head:]] = (fill_t[rk] + rnd)[seq]
        return rank_of, slot_of

    def nodes_of(self, rank: int) -> Tensor:
        """node ids owned by ``rank`` in slot order"""
        mine = torch.nonzero(self.rank_of == rank).flatten()
        return mine[torch.argsort(self.slot_of[mine])]

    def shard_rows(self, full: Tensor, rank: int) -> Tensor:
        """[N, ...] -> [cap, ...] rows of ``rank`` (zero padded)"""
        nodes = self.nodes_of(rank)
        out = full.new_zeros((self.cap,) + tuple(full.shape[1:]))
        out[: nodes.numel()] = full[nodes]
        return out

    def unshard_rows(self, gathered: Tensor) -> Tensor:
        """[P*cap, ...] gathered layout -> [N, ...] in node order"""
        return gathered[self.pid.to(gathered.device)]


class _PartitionTensors:
    def __init__(self, rank_of, slot_of, pid):
        self.rank_of, self.slot_of, self.pid = rank_of, slot_of, pid


# ------------------------------------------------------------------------------------------
# what one rank holds of the static graph
# ------------------------------------------------------------------------------------------
class HaloPlan:
    """The fixed all-to-all-v of one direction: which own rows go to which peer, and where the rows
    received from each peer sit in this rank's local row table ``[own rows (cap) | halo rows]``.
    ``reader`` / ``read`` : per edge, the node whose owner gathers and the node whose row it reads.
    Every rank holds the whole (static) edge list, so both sides of every pair are computed locally -
    no handshake; sender and receiver order a pair's rows by the owner's slot."""

    def __init__(self, reader: Tensor, read: Tensor, part: NodePartition, rank: int, device):
        """``reader`` / ``read`` live on the plan device (the GPU when there is one); so does everything below"""
        world, n = part.world, part.num_nodes
        pt = part.on(reader.device)
        self._pt, self._cap = pt, part.cap
        r_reader, r_read = pt.rank_of[reader], pt.rank_of[read]
        cross = r_reader != r_read
        # rows this rank receives: the distinct nodes its own readers read from other ranks
        mine = cross & (r_reader == rank)
        remote = torch.unique(read[mine])
        remote = remote[torch.argsort(pt.pid[remote])]                   # by owner, then by the owner's slot
        self.halo_nodes, self.num_halo = remote, int(remote.numel())
        self.recv_splits = torch.bincount(pt.rank_of[remote], minlength=world).tolist()
        # rows this rank sends: its own nodes that readers of other ranks read, per reading rank
        theirs = cross & (r_read == rank)
        pair = torch.unique(r_reader[theirs] * n + read[theirs])         # (reading rank, own node), distinct
        to_rank, node = pair // n, pair % n
        order = torch.argsort(to_rank * (part.cap + 1) + pt.slot_of[node])
        self.send_splits = torch.bincount(to_rank, minlength=world).tolist()
        self.send_slots = pt.slot_of[node][order].to(device)
        self.num_send = int(pair.numel())
        self._lut = None

    def local_index(self, part: NodePartition, rank: int, nodes: Tensor) -> Tensor:
        """node id -> row of the local table (own slot, or cap + position among the halo rows)"""
        if self._lut is None:
            pt = self._pt
            lut = torch.full((part.num_nodes,), -1, dtype=torch.int64, device=pt.rank_of.device)
            own = torch.nonzero(pt.rank_of == rank).flatten()
            lut[own] = pt.slot_of[own]
            lut[self.halo_nodes] = part.cap + torch.arange(self.num_halo, device=lut.device)
            self._lut = lut
        out = self._lut[nodes]
        if out.numel() and int(out.min()) < 0:
            raise RuntimeError("halo plan does not cover an edge endpoint")
        return out

    def emulate(self, own: Tensor, full: Tensor) -> Tensor:
        """the local table a real exchange would produce, built from the full [N, d] tensor (single-process
        checks of a shard: tests)"""
        return torch.cat([own, full[self.halo_nodes.to(full.device)]])


class RankShard:
    """What one rank holds of the static graph: the bucketed structures of its rows, in the rank's LOCAL
    row space ``[own rows (cap) | halo rows]``, and the two halo plans."""

    MIN_SPLIT_ROWS = 32      # fewer interior (or boundary) rows than this: one gather / transform over all rows

    def __init__(self, part: NodePartition, edge_index: Tensor, edge_type: Tensor, num_relations: int,
                 rank: int, device, backend, scheme: str = "pull", split: Optional[bool] = None):
        if scheme not in SCHEMES:
            raise ValueError(f"scheme must be one of {SCHEMES}, got {scheme!r}")
        device = torch.device(device)
        pd = device if device.type == "cuda" else torch.device("cpu")    # where the plans are computed
        ei, et = edge_index.to(pd), edge_type.to(pd)
        pt = part.on(pd)
        n, r = part.num_nodes, num_relations
        src, dst = ei[0], ei[1]
        cnt = torch.bincount(dst * r + et, minlength=n * r).clamp(min=1).to(torch.float32)
        self.part, self.rank, self.num_relations, self.scheme = part, rank, r, scheme
        self.cap, self.rows_all = part.cap, part.cap * part.world
        self.num_own = int(part.counts[rank])
        self.num_interior = int(part.num_interior[rank])
        m_in = pt.rank_of[dst] == rank               # in-edges of own rows, column order kept
        m_out = pt.rank_of[src] == rank              # out-edges of own rows
        self.num_in_edges, self.num_out_edges = int(m_in.sum()), int(m_out.sum())
        w_out = (1.0 / cnt[dst[m_out] * r + et[m_out]]).to(torch.float32)
        n_int = self.num_interior
        if split is None:
            split = True
        # interior / boundary halves (pull): worth two launches each only when both halves are real
        self.split = bool(split and scheme == "pull" and part.world > 1 and n_int >= self.MIN_SPLIT_ROWS
                          and self.cap - n_int >= self.MIN_SPLIT_ROWS)
        to_dev = lambda t: t.to(device)                                  # noqa: E731
        # backward of both schemes: out-edges of own rows over the local table [own | halo of g]
        self.halo_out = HaloPlan(src, dst, part, rank, device)          # a node's owner reads g of its out-neighbours
        key_out = pt.slot_of[src[m_out]]
        oth_out = self.halo_out.local_index(part, rank, dst[m_out])
        self.g_out = backend.make_shard(to_dev(key_out), to_dev(oth_out), to_dev(et[m_out]), self.cap,
                                        self.cap + self.halo_out.num_halo, r, to_dev(w_out))
        self.g_out_int = self.g_out_bnd = self.g_in_int = self.g_in_bnd = None
        if self.split:
            i_ = key_out < n_int                     # an interior row's out-neighbours are all own rows
            self.g_out_int = backend.make_shard(to_dev(key_out[i_]), to_dev(oth_out[i_]), to_dev(et[m_out][i_]), n_int,
                                                self.cap, r, to_dev(w_out[i_]))
            self.g_out_bnd = backend.make_shard(to_dev(key_out[~i_] - n_int), to_dev(oth_out[~i_]), to_dev(et[m_out][~i_]),
                                                self.cap - n_int, self.cap + self.halo_out.num_halo, r, to_dev(w_out[~i_]))
        if scheme == "pull":                         # forward: in-edges of own rows over [own | halo of x]
            self.halo_in = HaloPlan(dst, src, part, rank, device)       # a node's owner reads x of its in-neighbours
            key_in = pt.slot_of[dst[m_in]]
            oth_in = self.halo_in.local_index(part, rank, src[m_in])
            self.g_in = backend.make_shard(to_dev(key_in), to_dev(oth_in), to_dev(et[m_in]), self.cap,
                                           self.cap + self.halo_in.num_halo, r)
            if self.split:
                i_ = key_in < n_int
                self.g_in_int = backend.make_shard(to_dev(key_in[i_]), to_dev(oth_in[i_]), to_dev(et[m_in][i_]), n_int,
                                                   self.cap, r)
                self.g_in_bnd = backend.make_shard(to_dev(key_in[~i_] - n_int), to_dev(oth_in[~i_]), to_dev(et[m_in][~i_]),
                                                   self.cap - n_int, self.cap + self.halo_in.num_halo, r)
            self.g_push = None
            self.push_rows = None
        else:                                        # forward: partial sums of ALL rows from own sources
            self.halo_in = None
            self.g_in = None
            self.g_push = backend.make_shard(to_dev(pt.pid[dst[m_out]]), to_dev(pt.slot_of[src[m_out]]),
                                             to_dev(et[m_out]), self.rows_all, self.cap, r, to_dev(w_out))
            # the rows that receive a message from ANOTHER rank's source: only their partial outputs travel.  Per
            # owner, in slot order, padded to the longest list (reduce-scatter wants equal chunks); every rank
            # derives the same lists from the replicated edge list.
            remote_dst = torch.unique(dst[pt.rank_of[src] != pt.rank_of[dst]])
            remote_dst = remote_dst[torch.argsort(pt.pid[remote_dst])]
            per_owner = torch.bincount(pt.rank_of[remote_dst], minlength=part.world)
            width = int(per_owner.max()) if remote_dst.numel() else 0
            start = torch.cumsum(per_owner, 0) - per_owner
            pos = torch.arange(remote_dst.numel(), device=pd) - start[pt.rank_of[remote_dst]]
            send = torch.full((part.world * max(width, 1),), -1, dtype=torch.int64, device=pd)
            send[pt.rank_of[remote_dst] * max(width, 1) + pos] = pt.pid[remote_dst]
            own_rows = remote_dst[pt.rank_of[remote_dst] == rank]
            self.push_rows = {"width": max(width, 1), "send_pid": to_dev(send.clamp(min=0)), "send_valid": to_dev(send >= 0),
                              "own_slots": to_dev(pt.slot_of[own_rows]), "count": int(own_rows.numel()),
                              "total": int(remote_dst.numel())}
        remote_rows = n - self.num_own
        self.halo_fraction_in = (self.halo_in.num_halo / max(remote_rows, 1)) if self.halo_in else None
        self.halo_fraction_out = self.halo_out.num_halo / max(remote_rows, 1)


# ------------------------------------------------------------------------------------------
# exchanges
# ------------------------------------------------------------------------------------------
def _host_staged(t: Tensor, group) -> bool:
    # test rigs only (several ranks sharing one GPU, where RCCL refuses to run): gloo has no device
    # collectives, so the exchange is staged through the host
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Halo:
    """Exchange of the rows a rank's edges read: issue it, launch the work that only needs this rank's
    rows, then ``.table()`` = ``[own rows | received rows]`` when the gather needs it."""

    def __init__(self, own: Tensor, plan: HaloPlan, group):
        self.work = None
        d = own.size(1)
        self.tbl = own.new_empty(own.size(0) + plan.num_halo, d)
        self.tbl[: own.size(0)].copy_(own)
        send = own.index_select(0, plan.send_slots) if plan.num_send else own.new_empty(0, d)
        recv = self.tbl[own.size(0):]
        if dist.get_world_size(group) == 1:
            return
        if _host_staged(own, group):
            send_h = send.detach().cpu()
            recv_h = send_h.new_empty(plan.num_halo, d)
            dist.all_to_all_single(recv_h, send_h, plan.recv_splits, plan.send_splits, group=group)
            recv.copy_(recv_h)
            return
        self.work = dist.all_to_all_single(recv, send.contiguous(), plan.recv_splits, plan.send_splits, group=group,
                                           async_op=True)
        self._keep = send                        # alive until the collective has consumed it

    def table(self) -> Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.tbl


def _all_gather_rows(own: Tensor, world: int, group) -> Tensor:
    shape = (own.size(0) * world,) + tuple(own.shape[1:])
    if _host_staged(own, group):
        host = own.detach().contiguous().cpu()
        out_h = host.new_empty(shape)
        dist.all_gather_into_tensor(out_h, host, group=group)
        return out_h.to(own.device)
    out = own.new_empty(shape)
    dist.all_gather_into_tensor(out, own.contiguous(), group=group)
    return out


def _reduce_scatter_rows(full: Tensor, world: int, group) -> Tensor:
    """[P*cap, d] partial rows of every rank -> [cap, d] sums of this rank's rows"""
    cap = full.size(0) // world
    if _host_staged(full, group) or dist.get_backend(group) == "gloo":
        # gloo has no reduce-scatter: all-reduce, keep the own slab (test rigs only)
        host = full.detach().contiguous().cpu() if full.is_cuda else full.detach().clone()
        dist.all_reduce(host, group=group)
        k = dist.get_rank(group)
        return host[k * cap:(k + 1) * cap].to(full.device)
    out = full.new_empty((cap,) + tuple(full.shape[1:]))
    dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


def _flat_all_reduce(parts, group):
    """one flat all-reduce for a layer's parameter-gradient partial sums -> (work, flat, parts)"""
    parts = [t for t in parts if t is not None]
    flat = torch.cat([t.reshape(-1) for t in parts])
    if _host_staged(flat, group):
        host = flat.cpu()
        dist.all_reduce(host, group=group)
        flat.copy_(host)
        return None, flat, parts
    return dist.all_reduce(flat, group=group, async_op=True), flat, parts


def _unflatten(flat, parts, has_root, has_bias):
    outs, off = [], 0
    for t in parts:
        outs.append(flat[off: off + t.numel()].view_as(t))
        off += t.numel()
    it = iter(outs)
    return next(it), (next(it) if has_root else None), (next(it) if has_bias else None)


def _wait(red):
    if red[0] is not None:
        red[0].wait()


def _zero_pad_rows(g: Tensor, num_own: int) -> Tensor:
    """the slots past a rank's last node are padding: whatever arrives there must not reach a parameter gradient"""
    if num_own >= g.size(0):
        return g.contiguous()
    g = g.contiguous().clone()                  # (clone() alone keeps a transposed layout)
    g[num_own:] = 0
    return g


# ------------------------------------------------------------------------------------------
# autograd nodes
# ------------------------------------------------------------------------------------------
def _layer_fwd(x, weight, root, bias, relu, shard: RankShard, backend, group):
    """one layer on this rank's rows -> (out, agg as the parameter-gradient GEMM needs it)"""
    if shard.scheme == "pull":
        halo = _Halo(x, shard.halo_in, group)                              # in flight ...
        native = hasattr(backend, "layer_fwd")                             # whole segments as one native call each
        if not shard.split:
            tbl = halo.table()
            if native:
                return backend.layer_fwd(shard.g_in, tbl, x, weight, root, bias, relu)
            agg = backend.aggregate(shard.g_in, tbl)
            return backend.transform_fwd(agg, x, weight, root, bias, relu, shard.g_in, table=tbl), agg
        # ... behind the interior rows, which read own rows only; the boundary rows follow the wait.  Both halves
        # write row ranges of the same two tensors.
        k = shard.num_interior
        agg = x.new_empty(shard.cap, shard.num_relations * x.size(1))
        out = x.new_empty(shard.cap, weight.size(2))
        if native:
            backend.layer_fwd_rows(shard.g_in_int, x, x, weight, root, bias, relu, agg, out, 0, k)
            backend.layer_fwd_rows(shard.g_in_bnd, halo.table(), x, weight, root, bias, relu, agg, out, k, shard.cap)
            return out, agg
        backend.aggregate(shard.g_in_int, x, out=agg[:k])
        backend.transform_fwd(agg[:k], x[:k], weight, root, bias, relu, shard.g_in_int, table=x, out=out[:k])
        tbl = halo.table()
        backend.aggregate(shard.g_in_bnd, tbl, out=agg[k:])
        backend.transform_fwd(agg[k:], x[k:], weight, root, bias, relu, shard.g_in_bnd, table=tbl, out=out[k:])
        return out, agg
    # push: partial aggregates of ALL rows from own sources; the transform is linear, so partial outputs add
    part = backend.aggregate(shard.g_push, x)                                         # [P*cap, R*d_in]
    dummy = x.new_zeros(part.size(0), x.size(1))
    partial_out = backend.transform_fwd(part, dummy, weight, None, None, False, shard.g_push, table=x)
    world, cap, rank = shard.part.world, shard.cap, shard.rank
    out = partial_out[rank * cap:(rank + 1) * cap].clone()          # rows no other rank's source reaches are final here
    pr = shard.push_rows
    if world > 1 and pr["total"] > 0:
        # only the rows with a message from another rank's source travel: [P * width, d_out] -> this rank's [width, d_out]
        send = partial_out.index_select(0, pr["send_pid"]) * pr["send_valid"].view(-1, 1).to(partial_out.dtype)
        summed = _reduce_scatter_rows(send, world, group)
        out[pr["own_slots"]] = summed[: pr["count"]]
    if root is not None:
        out = out + x @ root
    if bias is not None:
        out = out + bias
    return (torch.relu(out) if relu else out), part


def _param_grads(agg, x, g, shard: RankShard, backend, group, has_root, has_bias):
    """parameter-gradient partial sums of this rank -> pending flat all-reduce"""
    r = shard.num_relations
    if shard.scheme == "pull":
        return _flat_all_reduce(backend.transform_bwd_params(agg, x, g, r, has_root, has_bias, shard.g_in), group)
    # push: grad_W from the partial aggregates of all rows against the gathered g; root / bias from own rows
    g_all = _all_gather_rows(g, shard.part.world, group)
    dummy = x.new_zeros(agg.size(0), x.size(1))
    gw, _, _ = backend.transform_bwd_params(agg, dummy, g_all, r, False, False, shard.g_push)
    return _flat_all_reduce([gw, x.t() @ g if has_root else None, g.sum(0) if has_bias else None], group)


def _input_grad(g, weight, root, relu_mask, shard: RankShard, backend, group, halo=None):
    """``halo``: the exchange of g's rows if the caller issued it already (so that work that needs own rows only -
    the parameter-gradient GEMM - runs while it is in flight)"""
    halo = halo if halo is not None else _Halo(g, shard.halo_out, group)
    native = hasattr(backend, "input_grad")
    if not shard.split:
        tbl = halo.table()
        if native:
            return backend.input_grad(shard.g_out, tbl, g, weight, root, relu_mask)
        gagg = backend.aggregate(shard.g_out, tbl)
        return backend.transform_bwd_input(gagg, g, weight, root, relu_mask, shard.g_out, table=tbl)
    k = shard.num_interior
    gagg = g.new_empty(shard.cap, shard.num_relations * g.size(1))
    if native:
        gx_i = backend.input_grad_rows(shard.g_out_int, g, g, weight, root, relu_mask, gagg, 0, k)
        gx_b = backend.input_grad_rows(shard.g_out_bnd, halo.table(), g, weight, root, relu_mask, gagg, k, shard.cap)
        return torch.cat([gx_i, gx_b])
    gx = g.new_empty(shard.cap, weight.size(1))
    mask_i = relu_mask[:k] if relu_mask is not None else None
    mask_b = relu_mask[k:] if relu_mask is not None else None
    backend.aggregate(shard.g_out_int, g, out=gagg[:k])
    backend.transform_bwd_input(gagg[:k], g[:k], weight, root, mask_i, shard.g_out_int, table=g, out=gx[:k])
    tbl = halo.table()
    backend.aggregate(shard.g_out_bnd, tbl, out=gagg[k:])
    backend.transform_bwd_input(gagg[k:], g[k:], weight, root, mask_b, shard.g_out_bnd, table=tbl, out=gx[k:])
    return gx


def _backward_layer(agg, x, g, weight, root, relu_mask, shard, backend, group, has_root, has_bias, need_x=True):
    """one layer's backward: the halo exchange of g first, the parameter-gradient GEMM (own rows only) and its
    all-reduce behind it, then the input gradient -> (pending reduction, grad_x | None)"""
    halo = _Halo(g, shard.halo_out, group) if need_x else None
    red = _param_grads(agg, x, g, shard, backend, group, has_root, has_bias)
    gx = _input_grad(g, weight, root, relu_mask, shard, backend, group, halo) if need_x else None
    return red, gx


def _encoder2_fwd(x, w1, root1, b1, w2, root2, b2, shard: RankShard, backend, group):
    """conv1 -> ReLU -> conv2 on this rank's rows -> (out, what the backward needs)"""
    x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
    h, agg1 = _layer_fwd(x, w1, root1, b1, True, shard, backend, group)
    out, agg2 = _layer_fwd(h, w2, root2, b2, False, shard, backend, group)
    return out, (x, agg1, h, agg2, w1, root1, w2, root2)


def _encoder2_bwd(saved, flags, g, shard: RankShard, backend, group, need_x: bool):
    x, agg1, h, agg2, w1, root1, w2, root2 = saved
    has_root1, has_b1, has_root2, has_b2 = flags
    g = _zero_pad_rows(g, shard.num_own)
    # per layer: halo exchange of the gradient rows, then the parameter-gradient GEMM + its all-reduce behind it,
    # then the input gradient (ReLU backward in its epilogue)
    red2, gz = _backward_layer(agg2, h, g, w2, root2, h, shard, backend, group, has_root2, has_b2)
    red1, gx = _backward_layer(agg1, x, gz, w1, root1, None, shard, backend, group, has_root1, has_b1, need_x=need_x)
    _wait(red2)
    _wait(red1)
    gw2, groot2, gb2 = _unflatten(red2[1], red2[2], has_root2, has_b2)
    gw1, groot1, gb1 = _unflatten(red1[1], red1[2], has_root1, has_b1)
    return gx, gw1, groot1, gb1, gw2, groot2, gb2


class _PartitionedEncoder2Function(torch.autograd.Function):
    """conv1 -> ReLU -> conv2 on this rank's rows as ONE autograd node (cf. conv._Encoder2Function):
    four exchanges per step, ReLU and its backward in the GEMM epilogues, parameter-gradient all-reduces
    in flight behind the gathers."""

    @staticmethod
    def forward(ctx, x, w1, root1, b1, w2, root2, b2, shard: RankShard, backend, group):
        out, saved = _encoder2_fwd(x, w1, root1, b1, w2, root2, b2, shard, backend, group)
        ctx.shard, ctx.backend, ctx.group = shard, backend, group
        ctx.flags = (root1 is not None, b1 is not None, root2 is not None, b2 is not None)
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, g):
        grads = _encoder2_bwd(ctx.saved_tensors, ctx.flags, g, ctx.shard, ctx.backend, ctx.group, ctx.needs_input_grad[0])
        return grads + (None, None, None)


class _PartitionedConvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_own, weight, root, bias, shard: RankShard, backend, group, relu=False):
        x_own, weight = x_own.contiguous(), weight.contiguous()
        out, agg = _layer_fwd(x_own, weight, root, bias, relu, shard, backend, group)
        ctx.shard, ctx.backend, ctx.group, ctx.relu = shard, backend, group, relu
        ctx.has_root, ctx.has_bias = root is not None, bias is not None
        ctx.save_for_backward(x_own, agg, weight, root, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g_own):
        x_own, agg, weight, root, out = ctx.saved_tensors
        shard, backend, group = ctx.shard, ctx.backend, ctx.group
        if ctx.relu:
            g_own = g_own * (out > 0)                                       # ReLU backward
        g_own = _zero_pad_rows(g_own, shard.num_own)
        red, gx = _backward_layer(agg, x_own, g_own, weight, root, None, shard, backend, group, ctx.has_root,
                                  ctx.has_bias, need_x=ctx.needs_input_grad[0])
        _wait(red)
        gw, groot, gbias = _unflatten(red[1], red[2], ctx.has_root, ctx.has_bias)
        return gx, gw, groot, gbias, None, None, None, None


def partitioned_conv(x_own: Tensor, weight: Tensor, root: Optional[Tensor], bias: Optional[Tensor],
                     shard: RankShard, backend, group=None, relu: bool = False) -> Tensor:
    """One R-GCN layer on this rank's rows (``[cap, d_in] -> [cap, d_out]``), optionally with
    the following ReLU fused into the transform."""
    return _PartitionedConvFunction.apply(x_own, weight, root, bias, shard, backend, group, relu)


class PartitionedEncoder:
    """conv1 -> relu -> conv2 over a node-partitioned graph (what ``bench.py --gpus N`` times).

    ``convs``: two ``RGCNConv``-like modules (``effective_weight()``, ``root``, ``bias``)
    with identical parameters on every rank; ``emb_full``: the [N, d] input table."""

    def __init__(self, edge_index: Tensor, edge_type: Tensor, num_nodes: int, num_relations: int,
                 emb_full: Tensor, convs: Sequence[torch.nn.Module], device, backend=None, group=None,
                 scheme: Optional[str] = None, assignment: Optional[Tensor] = None, split: Optional[bool] = None,
                 partition: str = "auto"):
        """``assignment`` (int64 [N], optional): node -> rank from a partitioner of the caller's, identical on every
        rank; else ``partition``: "auto" (default: the locality-aware ``NodePartition.clustered`` on degree-skewed
        graphs of PrimeKG's size, the degree-balanced deal otherwise), "clustered" or "deal" - computed on rank 0
        and broadcast; ``split``: interior / boundary overlap of the "pull" scheme (default on where a rank has >= 32
        rows of each kind)."""
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.group = group
        self.scheme = scheme or os.environ.get("RGCN_DIST_SCHEME", "pull")
        self.backend = backend if backend is not None else HipBackend()
        if assignment is not None:
            self.part = NodePartition.from_assignment(assignment, edge_index, self.world)
        else:
            self.part = NodePartition.shared(edge_index, num_nodes, group, device, method=partition)
        self.shard = RankShard(self.part, edge_index, edge_type, num_relations, self.rank, device,
                               self.backend, self.scheme, split=split)
        self.emb = self.part.shard_rows(emb_full, self.rank).to(device).requires_grad_(True)
        self.convs: List[torch.nn.Module] = [c.to(device) for c in convs]
        self.params = [self.emb] + [p for c in self.convs for p in c.parameters()]

    def exchange_summary(self) -> dict:
        """rows this rank receives per exchange, as a fraction of the rows it does not own"""
        s = self.shard
        return {"scheme": self.scheme, "partition": getattr(self.part, "method", "given"), "rows_own": s.num_own,
                "rows_remote": self.part.num_nodes - s.num_own,
                "rows_interior": s.num_interior, "interior_boundary_split": s.split,
                "push_rows_exchanged": (s.push_rows["total"] if s.push_rows else None),
                "halo_rows_forward": s.halo_in.num_halo if s.halo_in else None,
                "halo_fraction_forward": s.halo_fraction_in,
                "halo_rows_backward": s.halo_out.num_halo, "halo_fraction_backward": s.halo_fraction_out}

    def shard_rows(self, full: Tensor) -> Tensor:
        return self.part.shard_rows(full, self.rank)

    def forward(self) -> Tensor:
        c1, c2 = self.convs
        return _PartitionedEncoder2Function.apply(self.emb, c1.effective_weight(), c1.root, c1.bias,
                                                  c2.effective_weight(), c2.root, c2.bias, self.shard,
                                                  self.backend, self.group)

    def forward_layers(self) -> Tensor:
        """same result through the two per-layer autograd nodes (general API; used by tests)"""
        c1, c2 = self.convs
        h = partitioned_conv(self.emb, c1.effective_weight(), c1.root, c1.bias, self.shard, self.backend,
                             self.group, relu=True)
        return partitioned_conv(h, c2.effective_weight(), c2.root, c2.bias, self.shard, self.backend,
                                self.group)

    def step(self, cot_own: Tensor, explicit: bool = True) -> Tensor:
        """forward + backward with the given cotangent rows; grads land in ``.grad``.  ``explicit`` (default): the two
        halves are called directly, without the autograd engine - at N > 1 a rank's kernels shrink with 1 / N while the
        host cost of a step does not, and the engine's hop to its backward thread is ~100 us of it
        (``tools/host_profile.py``); plain-weight layers only (a basis-decomposed layer takes the autograd route)."""
        c1, c2 = self.convs
        if explicit and c1.num_bases is None and c2.num_bases is None:
            with torch.no_grad():
                flags = (c1.root is not None, c1.bias is not None, c2.root is not None, c2.bias is not None)
                out, saved = _encoder2_fwd(self.emb, c1.weight, c1.root, c1.bias, c2.weight, c2.root, c2.bias, self.shard,
                                           self.backend, self.group)
                grads = _encoder2_bwd(saved, flags, cot_own, self.shard, self.backend, self.group, True)
            targets = (self.emb, c1.weight, c1.root, c1.bias, c2.weight, c2.root, c2.bias)
            for p, gr in zip(targets, grads):
                if p is not None:
                    p.grad = gr
            return out
        out = self.forward()
        for p in self.params:
            p.grad = None
        out.backward(cot_own)
        return out

    def gather_output(self, own: Tensor) -> Tensor:
        """all ranks' rows -> [N, d] in node order (for checks)"""
        return self.part.unshard_rows(_all_gather_rows(own.detach(), self.world, self.group))


class ReplicatedEncoder:
    """Batch-replica mode ("alternative worth measuring"): every GPU holds
    the whole bucketed graph and runs the whole encoder for ITS mini-batch - the reference
    re-runs the full-graph encoder for each 1,024-edge batch anyway (``train.py:291-297``), so
    N GPUs work on N batches at once - and the gradients of all parameters including the
    embedding table (8.3 MB at C2) are averaged with ONE flat all-reduce per step.  No exchange
    inside the layer: per-GPU work is fixed as N grows (weak scaling).

    ``encoder_fn(emb, edge_index, edge_type, conv1, conv2) -> [N, d_out]`` is injectable for
    the CPU (gloo) tests; the product path is ``rgcn_encoder2`` on the HIP library."""

    def __init__(self, edge_index: Tensor, edge_type: Tensor, num_nodes: int, num_relations: int,
                 emb_full: Tensor, convs: Sequence[torch.nn.Module], device, encoder_fn=None, group=None):
        self.world = dist.get_world_size(group)
        self.group = group
        if encoder_fn is None:
            from .conv import rgcn_encoder2
            from . import ops
            encoder_fn = rgcn_encoder2
            self.edge_index, self.edge_type = edge_index.to(device), edge_type.to(device)
            ops.bucket(self.edge_index, self.edge_type, num_nodes, num_relations)
        else:
            self.edge_index, self.edge_type = edge_index.to(device), edge_type.to(device)
        self.encoder_fn = encoder_fn
        self.emb = emb_full.to(device).clone().requires_grad_(True)
        self.convs: List[torch.nn.Module] = [c.to(device) for c in convs]
        self.params = [self.emb] + [p for c in self.convs for p in c.parameters()]
        self._flat = torch.empty(sum(p.numel() for p in self.params), device=device, dtype=self.emb.dtype)
        self._views, off = [], 0
        for p in self.params:
            self._views.append(self._flat[off: off + p.numel()].view_as(p))
            off += p.numel()

    def step(self, cot: Tensor) -> Tensor:
        """forward + backward for this rank's cotangent, then the cross-rank mean of every
        parameter gradient (left in ``.grad``, identical on all ranks)."""
        out = self.encoder_fn(self.emb, self.edge_index, self.edge_type, self.convs[0], self.convs[1])
        for p in self.params:
            p.grad = None
        out.backward(cot)
        torch._foreach_copy_(self._views, [p.grad for p in self.params])
        if _host_staged(self._flat, self.group):
            host = self._flat.cpu()
            dist.all_reduce(host, group=self.group)
            self._flat.copy_(host)
        else:
            dist.all_reduce(self._flat, group=self.group)
        self._flat.div_(self.world)
        for p, v in zip(self.params, self._views):
            p.grad = v
        return out

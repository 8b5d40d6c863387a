"""Rows a rank receives per halo exchange as a fraction of the rows it does not own (dist.RankShard), for the
C2 (PrimeKG-shaped) and C4 (uniform 500k / 20M / 16) graphs at P = 2, 4, 8.  CPU only: python tools/halo_fraction.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import dist as rdist, synth  # noqa: E402


class _Rec:
    def make_shard(self, *a, **k):
        return a


for name, (ei, et, n, r) in (("C2", synth.primekg_like(seed=42)),
                            ("C4", synth.uniform_graph(500_000, 20_000_000, 16, seed=42))):
    for world in (2, 4, 8):
        part = rdist.NodePartition(ei, n, world)
        sh = rdist.RankShard(part, ei, et, r, 0, torch.device("cpu"), _Rec())
        deg = torch.bincount(ei[1], minlength=n).float()
        per = torch.zeros(world).index_add_(0, part.rank_of, deg)
        print(f"{name} P={world}: own rows {sh.num_own}, halo rows fwd {sh.halo_in.num_halo} "
              f"({sh.halo_fraction_in:.3f} of the remote rows), bwd {sh.halo_out.num_halo} ({sh.halo_fraction_out:.3f}); "
              f"in-edge balance max/mean {float(per.max() / per.mean()):.4f}")

"""Round 4 probe: does the parameter-gradient GEMM (planes kernel: 96 KB of LDS, 8 waves per CU, bound by its LDS-DMA
stream) run BESIDE the transposed gather that does not depend on it (bound by indexed L2 reads, 4 KB of LDS per
workgroup)?  C2's graph and shapes; each variant captured as a HIP graph of `reps` repetitions and replayed.

    (needs the plane entry points of commit e229a4a)  python tools/overlap_probe2.py          ->  us per (TN + gather) pair: one stream / two streams, both launch orders
"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primekg_rgcn_linkprediction_amd import ops, synth, _lib   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ei, et, n, r = synth.primekg_like()
    ei, et = ei.to(dev), et.to(dev)
    graph = ops.bucket(ei, et, n, r)
    lib = _lib.load()
    reps = 10
    for d_in, d_out, gather_d in ((128, 128, 128), (64, 128, 128), (64, 128, 64)):
        agg = torch.randn(n, r * d_in, device=dev) * 0.05
        x = torch.randn(n, d_in, device=dev) * 0.05
        g = torch.randn(n, d_out, device=dev) * 0.01
        table = torch.randn(n, gather_d, device=dev) * 0.01
        am = ops.amax_buffer(dev, 3)
        ops.absmax_many([agg, x, g], [am[0], am[1], am[2]])
        planes = {}
        for name, t, a in (("agg", agg, am[0]), ("x", x, am[1]), ("g", g, am[2])):
            hi = torch.empty(t.shape, dtype=torch.float16, device=dev)
            lo = torch.empty(t.shape, dtype=torch.float16, device=dev)
            _lib.check(lib.rgcn_split_planes(t.data_ptr(), t.numel(), a.data_ptr(), 1.0, hi.data_ptr(), lo.data_ptr(),
                                             ops._stream()), "rgcn_split_planes")
            planes[name] = (hi, lo)
        gw = torch.empty(r, d_in, d_out, device=dev)
        groot = torch.empty(d_in, d_out, device=dev)
        nbytes = lib.rgcn_transform_bwd_params_split_workspace_bytes(n, r, d_in, d_out)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        job = _lib.SlabJob()
        out = torch.empty(n, r * gather_d, device=dev)

        def tn():
            rc = lib.rgcn_transform_bwd_params_planes_begin(
                planes["agg"][0].data_ptr(), planes["agg"][1].data_ptr(), planes["x"][0].data_ptr(), planes["x"][1].data_ptr(),
                planes["g"][0].data_ptr(), planes["g"][1].data_ptr(), None, n, r, d_in, d_out, am[0].data_ptr(), 1.0,
                am[1].data_ptr(), am[2].data_ptr(), 0, gw.data_ptr(), groot.data_ptr(), None, None, 0, ws.data_ptr(), nbytes,
                ops._stream(), ctypes.byref(job))
            _lib.check(rc, "planes_begin")

        def tn_old():
            ops.transform_bwd_params(agg, x, g, r, want_bias=False, defer=True, amax=(am[0], am[1], am[2]))

        def gather():
            ops.aggregate(graph, table, transposed=True, out=out)

        def capture(body):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    body()
            return gr

        side = torch.cuda.Stream()

        def seq(first, second):
            def body():
                first()
                second()
            return body

        def par(main_fn, side_fn):
            def body():
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    side_fn()
                main_fn()
                cur.wait_stream(side)
            return body

        def timed(gr):
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                gr.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / (10 * reps) * 1e3

        variants = [("TN(planes) alone", lambda: tn()), ("TN(coop, fp32 in) alone", lambda: tn_old()), ("gather alone", lambda: gather()),
                    ("TN(planes) -> gather, one stream", seq(tn, gather)),
                    ("gather on main || TN(planes) on side", par(gather, tn)),
                    ("TN(planes) on main || gather on side", par(tn, gather)),
                    ("TN(coop) -> gather, one stream", seq(tn_old, gather)),
                    ("gather on main || TN(coop) on side", par(gather, tn_old))]
        print(f"TN [{n} x {(r + 1) * d_in}]^T x [{n} x {d_out}]  beside the transposed gather of {gather_d}-wide rows")
        for name, body in variants:
            gr = capture(body)
            print(f"    {name:44s} {timed(gr):8.2f} us")


if __name__ == "__main__":
    main()

"""Diagnostic: how much of a transform launch is fixed cost (launch, prologue, epilogue)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primekg_rgcn_linkprediction_amd import ops

dev = torch.device("cuda:0")
N = 30926

def timeit(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for d_out in (64, 128):
    for R, d_in in ((1, 32), (1, 64), (1, 128), (3, 128), (7, 128)):
        agg = torch.randn(N, R * d_in, device=dev); x = torch.randn(N, d_in, device=dev)
        w = torch.randn(R, d_in, d_out, device=dev) * 0.1
        bias = torch.randn(d_out, device=dev)
        t = timeit(lambda: ops.transform_fwd(agg, x, w, None, bias))
        K = R * d_in
        print(f"fwd  N={d_out:4d} K={K:4d}: {t:7.2f} us   ({2.0*N*K*d_out/t/1e6:6.1f} TF)")
# plain elementwise store of the same output size for scale
y = torch.empty(N, 128, device=dev)
print(f"torch fill_ [30926x128]: {timeit(lambda: y.fill_(1.0)):7.2f} us")
print(f"torch empty kernel-ish (zero_ of 4 floats): {timeit(lambda: y[:1, :4].zero_()):7.2f} us")

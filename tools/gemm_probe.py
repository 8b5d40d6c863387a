"""Diagnostic: time the transform kernels in isolation (back-to-back launches, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primekg_rgcn_linkprediction_amd import ops

dev = torch.device("cuda:0")
N, R = 30926, 3
torch.manual_seed(0)

def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us

for d_in, d_out in ((64, 128), (128, 128)):
    agg = torch.randn(N, R * d_in, device=dev); x = torch.randn(N, d_in, device=dev)
    w = torch.randn(R, d_in, d_out, device=dev) * 0.1; root = torch.randn(d_in, d_out, device=dev) * 0.1
    bias = torch.randn(d_out, device=dev); g = torch.randn(N, d_out, device=dev)
    gagg = torch.randn(N, R * d_out, device=dev)
    flops = 2.0 * N * (R + 1) * d_in * d_out
    t = timeit(lambda: ops.transform_fwd(agg, x, w, root, bias))
    print(f"fwd   {d_in}->{d_out}: {t:7.2f} us  {flops / t / 1e6:6.1f} TF")
    t = timeit(lambda: ops.transform_bwd_input(gagg, g, w, root))
    print(f"bwd_x {d_in}->{d_out}: {t:7.2f} us  {flops / t / 1e6:6.1f} TF")
    t = timeit(lambda: ops.transform_bwd_params(agg, x, g, R))
    print(f"bwd_w {d_in}->{d_out}: {t:7.2f} us  {flops / t / 1e6:6.1f} TF (incl. slab reduce)")
# torch/rocBLAS reference point for the same forward GEMM shape (plain library GEMM, for scale only)
a = torch.randn(N, 512, device=dev); b = torch.randn(512, 128, device=dev)
t = timeit(lambda: a @ b)
print(f"torch.matmul [30926x512]x[512x128]: {t:7.2f} us  {2.0*N*512*128/t/1e6:6.1f} TF")
# library reference for the parameter-gradient shape: [512 x 30926] x [30926 x 128]
a = torch.randn(N, 512, device=dev); g2 = torch.randn(N, 128, device=dev)
t = timeit(lambda: a.t() @ g2)
print(f"torch.matmul A^T[512x30926] x G[30926x128]: {t:7.2f} us  {2.0*N*512*128/t/1e6:6.1f} TF")
a = torch.randn(N, 256, device=dev)
t = timeit(lambda: a.t() @ g2)
print(f"torch.matmul A^T[256x30926] x G[30926x128]: {t:7.2f} us  {2.0*N*256*128/t/1e6:6.1f} TF")

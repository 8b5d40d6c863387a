"""Diagnostic: wall time of one reference-protocol training epoch (full encoder fwd+bwd per 1,024-edge
batch, BCE, clip, Adam) on a PrimeKG-shaped synthetic graph of the true train-graph size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primekg_rgcn_linkprediction_amd import train as T

dev = torch.device("cuda:0")
tr, va, full, te = T.synthetic_data(num_edges=1_708_556, seed=42)    # 2 x 854,278 kg rows; ~1.68M train columns
args = T.parse_args(["--epochs", "1", "--output_dir", "/tmp/epoch_probe"] + sys.argv[1:])   # e.g. --no_hip_graph
torch.manual_seed(42)
trainer = T.Trainer(T.create_model(tr["num_nodes"], 3, args), tr, va, full, dev, args)
steps = (tr["edge_index"].size(1) + args.batch_size - 1) // args.batch_size
trainer.train_epoch(max_steps=20)                 # warm-up (bucketing, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
limit = int(os.environ.get("RGCN_EPOCH_STEPS", "0")) or None          # a shorter run for the profiler
loss, acc = trainer.train_epoch(max_steps=limit)
steps = min(steps, limit) if limit else steps
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"train columns {tr['edge_index'].size(1):,}  steps/epoch {steps}  epoch {t:.2f} s  "
      f"{t / steps * 1e3:.3f} ms/step  loss {loss:.4f} acc {acc:.4f}")
t0 = time.perf_counter()
vl, va_acc = trainer.validate()
torch.cuda.synchronize()
print(f"validate {time.perf_counter() - t0:.2f} s  val loss {vl:.4f} acc {va_acc:.4f}")
if args.no_hip_graph:                             # which passes of the eager step are issued natively (ops.Region)
    import gc
    import warnings
    from primekg_rgcn_linkprediction_amd import ops
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")           # (isinstance on every live object wakes deprecated module attributes)
        graphs = [o for o in gc.get_objects() if isinstance(o, ops.BucketedGraph)]
    for g in graphs:
        for (name, *_), state in getattr(g, "_regions", {}).items():
            what = "native" if isinstance(state, ops._Plan) else (state if isinstance(state, str) else state[0])
            print(f"  region {name}: {what}")

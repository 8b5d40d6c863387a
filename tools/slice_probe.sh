#!/bin/bash
# GPU box (PROBE build of rgcn_aggregate.hip): the gathers with their rows cut into column slices, one slice per XCD
# (RGCN_PROBE_SLICE_BYTES = 128 / 64), against the whole-row gathers: parity of a few tests, then bench lines
set -e
mkdir -p gpurun_out
for b in 0 128 64; do
  export RGCN_PROBE_SLICE_BYTES=$b
  python -m pytest tests/test_gpu_parity.py -x -q -k "config_c1 or aggregate or gather" > gpurun_out/r04w_tests_$b.log 2>&1 || { tail -20 gpurun_out/r04w_tests_$b.log; exit 1; }
  tail -1 gpurun_out/r04w_tests_$b.log
  for rep in 1 2; do
    python bench.py --steps 50 --warmup 10 --no-secondary > gpurun_out/r04w_bench_${b}_$rep.json 2> gpurun_out/r04w_bench_${b}_$rep.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r04w_bench_${b}_$rep.json').read().strip().splitlines()[-1])
print('slice bytes $b:', round(d['ms_per_step'],4), [(k['kernel'], round(k['avg_us'],1)) for k in d['gather_kernels']])
PY
  done
done

#!/bin/bash
# GPU box: the TN prologue change - stamps, the transform / determinism tests, a bench line
set -e
mkdir -p gpurun_out
tools/gemm_stamps > gpurun_out/r04t_stamps.txt 2>&1
grep -A3 "TN params" gpurun_out/r04t_stamps.txt || true
python -m pytest tests/test_gpu_parity.py -x -q -k "transform or params or config_c1 or config_c2 or determin or bitwise" > gpurun_out/r04t_tests.log 2>&1 || { tail -30 gpurun_out/r04t_tests.log; exit 1; }
tail -2 gpurun_out/r04t_tests.log
python bench.py --steps 50 --warmup 10 --no-secondary > gpurun_out/r04t_bench.json 2> gpurun_out/r04t_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04t_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], [(c['call'],c['K'],round(c['avg_us'],2)) for c in d['transform_calls']])
PY

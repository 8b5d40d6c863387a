#!/bin/bash
# GPU box: the regression test for fuzz case 777/109, then the two fuzz scripts on fresh seeds (progress lines, no pipes)
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "two_nodes_with_segments or config_c1" > gpurun_out/r04z_regress.log 2>&1 || { cat gpurun_out/r04z_regress.log; exit 1; }
echo "regression test ok"
timeout -k 10 500 python tools/fuzz_encoder.py 150 777 > gpurun_out/r04z_fuzz_777.txt 2>&1 || true
tail -n 3 gpurun_out/r04z_fuzz_777.txt
timeout -k 10 400 python tools/fuzz_encoder.py 120 4242 > gpurun_out/r04z_fuzz_4242.txt 2>&1 || true
tail -n 3 gpurun_out/r04z_fuzz_4242.txt

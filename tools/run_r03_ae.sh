#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03ae}
timeout -k 10 120 tools/gemm_stamps > $out/${tag}_stamps.txt 2>&1
grep -E "TN |NT |prologue|main loop|lifetime" $out/${tag}_stamps.txt
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_$i.json 2> $out/${tag}_$i.err || exit 1
  python3 - <<PY
import json
r = json.load(open("$out/${tag}_$i.json"))
print("run $i", r["ms_per_step"], r["roofline_mfma"]["sum_transform_us_per_step"])
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -o p -- python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_prof.json 2> $out/${tag}_prof.err
cp "$(find $out/prof_${tag} -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
grep -E "k_gemm_tn|k_gemm_nt|k_aggregate<32" $out/${tag}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "params or same_bits or c2_full" 2>&1 | tail -2

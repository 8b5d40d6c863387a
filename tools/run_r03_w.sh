#!/bin/bash
# the wide NT layout (RGCN_NT_LAYOUT=wide: 4 waves x 32 rows x 128 columns): GPU suite on it, stamps, bench A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03w}
RGCN_NT_LAYOUT=wide timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
for cfg in default wide; do
  echo "=== $cfg" >> $out/${tag}_stamps.txt
  if [ $cfg = wide ]; then RGCN_NT_LAYOUT=wide timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1
  else timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1; fi
done
grep -E "===|NT |main loop|prologue|epilogue|lifetime" $out/${tag}_stamps.txt
for i in 1 2 3; do
  for cfg in default wide; do
    if [ $cfg = wide ]; then export RGCN_NT_LAYOUT=wide; else unset RGCN_NT_LAYOUT; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_${cfg}_$i.json 2> $out/${tag}_${cfg}_$i.err || exit 1
    python3 - <<PY
import json
r = json.load(open("$out/${tag}_${cfg}_$i.json"))
print("$cfg run $i", r["ms_per_step"], r["roofline_mfma"]["sum_transform_us_per_step"])
PY
  done
done

#!/bin/bash
# A/B of the software-pipelined NT k loop (default) against round 2's loop (RGCN_NT_PIPE=0): parity tests first, then the
# stamp probe and the bench line, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03k}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -30 $out/${tag}_pytest.log; exit 1; }
tail -3 $out/${tag}_pytest.log
for cfg in "RGCN_NT_PIPE=1" "RGCN_NT_PIPE=0"; do
  echo "=== $cfg" >> $out/${tag}_stamps.txt
  env $cfg timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1
done
grep -E "===|NT |main loop|prologue|lifetime" $out/${tag}_stamps.txt
for i in 1 2 3; do
  for p in 1 0; do
    RGCN_NT_PIPE=$p timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_pipe${p}_$i.json 2> $out/${tag}_pipe${p}_$i.err || exit 1
    python3 - <<PY
import json
r = json.load(open("$out/${tag}_pipe${p}_$i.json"))
print("pipe=$p run $i", r["ms_per_step"], r["roofline_mfma"]["sum_transform_us_per_step"])
PY
  done
done
timeout -k 10 120 tools/overlap_probe > $out/${tag}_overlap.txt 2>&1; cat $out/${tag}_overlap.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03j}
for cfg in "default" "RGCN_NT_COLS=64"; do
  echo "=== $cfg" >> $out/${tag}_stamps.txt
  if [ "$cfg" = default ]; then timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1
  else env $cfg timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1; fi
done
grep -E "===|NT forward|main loop|prologue|lifetime" $out/${tag}_stamps.txt
bash tools/measure.sh r03 2 > $out/${tag}_measure2.log 2>&1; tail -5 $out/${tag}_measure2.log

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03d}
for cfg in "default" "RGCN_NT_ROWS=128" "RGCN_TN_KERNEL=split"; do
  echo "=== $cfg" >> $out/${tag}_stamps.txt
  if [ "$cfg" = default ]; then timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1
  else env $cfg timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1; fi
done
cat $out/${tag}_stamps.txt

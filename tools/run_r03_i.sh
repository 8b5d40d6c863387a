#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03i}
for cfg in "default" "RGCN_NT_ROWS=128"; do
  for bin in gemm_stamps gemm_stamps_nosplit; do
    echo "=== $cfg $bin" >> $out/${tag}_stamps.txt
    if [ "$cfg" = default ]; then timeout -k 10 120 tools/$bin >> $out/${tag}_stamps.txt 2>&1
    else env $cfg timeout -k 10 120 tools/$bin >> $out/${tag}_stamps.txt 2>&1; fi
  done
done
grep -E "===|NT forward|main loop" $out/${tag}_stamps.txt

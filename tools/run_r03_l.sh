#!/bin/bash
# depth experiment: the one-pass (fp16) NT kernel with 2 and with 3 k-tiles in flight at equal occupancy (2 workgroups per CU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03l}
for bin in gemm_stamps gemm_stamps_d3; do
  for half in 1 0; do
    echo "=== $bin half=$half" >> $out/${tag}_stamps.txt
    if [ $half = 1 ]; then RGCN_STAMP_HALF=1 timeout -k 10 120 tools/$bin >> $out/${tag}_stamps.txt 2>&1
    else timeout -k 10 120 tools/$bin >> $out/${tag}_stamps.txt 2>&1; fi
  done
done
grep -E "===|NT |main loop|prologue" $out/${tag}_stamps.txt

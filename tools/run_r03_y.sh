#!/bin/bash
# the ping-pong NT k loop (RGCN_NT_PP=1; used where no hub rows are left to the launch: RGCN_DEFER_HUBS=0 for the A/B)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03y}
export RGCN_DEFER_HUBS=0
RGCN_NT_PP=1 timeout -k 10 240 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transform or split or same_bits" > $out/${tag}_pytest1.log 2>&1 || { tail -30 $out/${tag}_pytest1.log; exit 1; }
tail -2 $out/${tag}_pytest1.log
RGCN_NT_PP=1 timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
for cfg in rows64 rows128 pingpong; do
  echo "=== $cfg" >> $out/${tag}_stamps.txt
  case $cfg in
    rows64) timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1 ;;
    rows128) RGCN_NT_ROWS=128 timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1 ;;
    pingpong) RGCN_NT_PP=1 timeout -k 10 120 tools/gemm_stamps >> $out/${tag}_stamps.txt 2>&1 ;;
  esac
done
grep -E "===|NT |main loop|lifetime" $out/${tag}_stamps.txt
for i in 1 2 3; do
  for cfg in rows64 pingpong; do
    if [ $cfg = pingpong ]; then export RGCN_NT_PP=1; else unset RGCN_NT_PP; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_${cfg}_$i.json 2> $out/${tag}_${cfg}_$i.err || exit 1
    python3 - <<PY
import json
r = json.load(open("$out/${tag}_${cfg}_$i.json"))
print("$cfg (no hub deferral) run $i", r["ms_per_step"], r["roofline_mfma"]["sum_transform_us_per_step"])
PY
  done
done

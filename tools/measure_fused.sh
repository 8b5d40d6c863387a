#!/bin/bash
# The no-grad encoder as one kernel per layer against gather -> transform: times (events + rocprofv3 kernel stats)
# and HBM-side bytes (FETCH_SIZE / WRITE_SIZE, one counter per pass) at C2 and at C4's graph on one GPU.
# usage: bash tools/measure_fused.sh <tag>      -> gpurun_out/<tag>_fused_*
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
python3 tools/fused_probe.py c2 both 8 16 32 > $out/${tag}_fused_c2.txt 2>&1
python3 tools/fused_probe.py c4 both 16 > $out/${tag}_fused_c4.txt 2>&1
for w in c2 c4; do
  for mode in plain fused; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_fused_${w}_$mode -o p -- python3 tools/fused_probe.py $w $mode 16 \
        > /dev/null 2> $out/${tag}_fused_prof.err
    cp "$(find $out/prof_${tag}_fused_${w}_$mode -name '*kernel_stats.csv' | head -1)" $out/${tag}_fused_${w}_${mode}_kernel_stats.csv
  done
done
for mode in plain fused; do
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_${tag}_fused_c4_$mode/$set -o p -- python3 tools/fused_probe.py c4 $mode 16 \
        > /dev/null 2> $out/${tag}_fused_pmc.err
  done
  python3 tools/pmc_summary.py $out/pmc_${tag}_fused_c4_$mode $out/${tag}_fused_c4_${mode}_pmc.json \
      "tools/fused_probe.py c4 $mode 16: no-grad 2-layer forward on C4's graph (500k nodes / 20M edges / 16 relations), averages per dispatch" \
      > $out/${tag}_fused_c4_${mode}_pmc.txt
done
cat $out/${tag}_fused_c2.txt $out/${tag}_fused_c4.txt $out/${tag}_fused_c4_plain_pmc.txt $out/${tag}_fused_c4_fused_pmc.txt

#!/bin/bash
# kernel statistics of the training step (reference protocol, default dropouts), graph replay, 300 steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03n}
export RGCN_EPOCH_STEPS=300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_train -o p -- python3 tools/epoch_time.py > $out/${tag}_train_prof.log 2>&1
cp "$(find $out/prof_${tag}_train -name '*kernel_stats.csv' | head -1)" $out/${tag}_train_kernel_stats.csv
grep -v amdgpu $out/${tag}_train_prof.log | tail -4
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/${tag}_train_kernel_stats.csv")))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
PY

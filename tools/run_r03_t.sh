#!/bin/bash
# what the riding slab reduction costs the transposed gathers: kernel statistics with the ride (default) and without
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03t}
for mode in rides norides; do
  if [ $mode = norides ]; then export RGCN_SLAB_RIDES=0; fi
  python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_${mode}.json 2> $out/${tag}_${mode}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$mode -o p -- python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_${mode}_prof.json 2> $out/${tag}_${mode}_prof.err
  cp "$(find $out/prof_${tag}_$mode -name '*kernel_stats.csv' | head -1)" $out/${tag}_${mode}_kernel_stats.csv
  python3 - <<PY
import csv, json
r = json.load(open("$out/${tag}_${mode}.json"))
print("== $mode", r["ms_per_step"])
for row in csv.DictReader(open("$out/${tag}_${mode}_kernel_stats.csv")):
    n = row["Name"]
    if any(k in n for k in ("k_aggregate", "k_slab", "k_reduce_partials", "k_gemm_tn")):
        print(f"  {n[:60]:60s} {row['Calls']:>5s} {float(row['AverageNs'])/1e3:7.2f} us")
PY
done

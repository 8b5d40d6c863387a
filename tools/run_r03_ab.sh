#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03ab}
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_$i.json 2> $out/${tag}_$i.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -o p -- python3 bench.py --no-cpu-baseline --no-secondary > $out/${tag}_prof.json 2> $out/${tag}_prof.err
cp "$(find $out/prof_${tag} -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
python3 - <<PY
import csv, json
for i in (1, 2, 3):
    r = json.load(open(f"$out/${tag}_{i}.json"))
    print("run", i, r["ms_per_step"])
for row in csv.DictReader(open("$out/${tag}_kernel_stats.csv")):
    n = row["Name"]
    if any(k in n for k in ("k_aggregate", "k_gemm_tn", "k_gemm_nt")):
        print(f"  {n[:62]:62s} {row['Calls']:>5s} {float(row['AverageNs'])/1e3:7.2f} us")
PY

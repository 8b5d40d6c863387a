#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03ac}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-secondary --workload c3 > $out/${tag}_c3_$i.json 2> $out/${tag}_c3_$i.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_c3 -o p -- python3 bench.py --no-cpu-baseline --no-secondary --workload c3 > $out/${tag}_c3_prof.json 2> $out/${tag}_c3_prof.err
cp "$(find $out/prof_${tag}_c3 -name '*kernel_stats.csv' | head -1)" $out/${tag}_c3_kernel_stats.csv
python3 - <<PY
import csv, json
for i in (1, 2):
    print("c3 run", i, json.load(open(f"$out/${tag}_c3_{i}.json"))["ms_per_step"])
for row in list(csv.DictReader(open("$out/${tag}_c3_kernel_stats.csv")))[:24]:
    if int(row["Calls"]) >= 50:
        print(f"  {row['Name'][:80]:80s} {row['Calls']:>5s} {float(row['AverageNs'])/1e3:7.2f} us")
PY

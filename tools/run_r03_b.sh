#!/bin/bash
# round 3, second GPU pass: the -m gpu tier on the new kernels, A/B of the cooperative parameter-gradient GEMM and of
# the native step, host profile, rocprofv3 kernel stats of the headline command.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03b}
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -8 $out/${tag}_pytest.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err; echo "$name rc=$?"; }
b coop
RGCN_TN_KERNEL=split b tnsplit
b coop2
RGCN_TN_KERNEL=split b tnsplit2
b eager_native --no-graph
RGCN_NATIVE_STEP=0 b eager_wrappers --no-graph
python3 tools/host_profile.py 12 > $out/${tag}_host_profile.txt 2>&1; head -3 $out/${tag}_host_profile.txt
RGCN_NATIVE_STEP=0 python3 tools/host_profile.py 12 > $out/${tag}_host_profile_wrappers.txt 2>&1; head -3 $out/${tag}_host_profile_wrappers.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_c2 -o p -- python3 bench.py --no-cpu-baseline --no-secondary \
    > $out/${tag}_c2_prof_bench.json 2> $out/${tag}_c2_prof.err
cp "$(find $out/prof_${tag}_c2 -name '*kernel_stats.csv' | head -1)" $out/${tag}_c2_kernel_stats.csv
python3 - <<PY
import json, csv
for n in ("coop", "tnsplit", "coop2", "tnsplit2", "eager_native", "eager_wrappers"):
    try:
        r = json.loads([l for l in open("$out/${tag}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(r["ms_per_step"], 4), r["config"]["launch"], [(c["call"], c["K"], round(c["avg_us"], 1)) for c in r["transform_calls"]])
    except Exception as exc:
        print(n, "unreadable:", exc)
for row in list(csv.DictReader(open("$out/${tag}_c2_kernel_stats.csv")))[:16]:
    print(row["Name"][:70], row["Calls"], row["AverageNs"])
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03c}
timeout -k 10 120 tools/gemm_stamps > $out/${tag}_stamps.txt 2>&1; echo "stamps rc=$?"; cat $out/${tag}_stamps.txt
timeout -k 10 700 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -8 $out/${tag}_pytest.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err; echo "$name rc=$?"; }
b wide
b wide2
python3 tools/host_profile.py 12 > $out/${tag}_host_profile.txt 2>&1; head -3 $out/${tag}_host_profile.txt
python3 - <<PY
import json
for n in ("wide", "wide2"):
    try:
        r = json.loads([l for l in open("$out/${tag}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(r["ms_per_step"], 4), r["config"]["launch"], [(c["call"], c["K"], round(c["avg_us"], 1)) for c in r["transform_calls"]])
    except Exception as exc:
        print(n, "unreadable:", exc)
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03h}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -8 $out/${tag}_pytest.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err; echo "$name rc=$?"; }
b rides
RGCN_PREP_RIDES=0 b norides
b rides2
RGCN_PREP_RIDES=0 b norides2
python3 tools/host_profile.py 10 > $out/${tag}_host_profile.txt 2>&1; head -6 $out/${tag}_host_profile.txt
python3 - <<PY
import json
for n in ("rides", "norides", "rides2", "norides2"):
    try:
        r = json.loads([l for l in open("$out/${tag}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(r["ms_per_step"], 4), r["config"]["launch"], [(k["kernel"], round(k["avg_us"], 1)) for k in r["gather_kernels"]])
    except Exception as exc:
        print(n, "unreadable:", exc)
PY

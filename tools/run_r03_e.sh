#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03e}
timeout -k 10 700 python3 -m pytest tests -m gpu -q -x > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -6 $out/${tag}_pytest.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err; echo "$name rc=$?"; }
RGCN_HOT_KB=32 b hot32
RGCN_HOT_KB=16 b hot16
RGCN_HOT_KB=0 b hot0
RGCN_HOT_KB=32 b hot32b
RGCN_HOT_KB=16 b hot16b
RGCN_HOT_KB=0 b hot0b
timeout -k 10 120 tools/gemm_stamps > $out/${tag}_stamps.txt 2>&1; grep -E "NT|TN|prologue" $out/${tag}_stamps.txt
python3 - <<PY
import json
for n in ("hot32", "hot16", "hot0", "hot32b", "hot16b", "hot0b"):
    try:
        r = json.loads([l for l in open("$out/${tag}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(r["ms_per_step"], 4), [(k["kernel"], round(k["avg_us"], 1)) for k in r["gather_kernels"]],
              [(c["call"], c["K"], round(c["avg_us"], 1)) for c in r["transform_calls"]])
    except Exception as exc:
        print(n, "unreadable:", exc)
PY

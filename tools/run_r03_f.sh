#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03f}
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hot_rows or aggregate or c2_full or deferred" > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -4 $out/${tag}_pytest.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" > $out/${tag}_${name}.json 2> $out/${tag}_${name}.err; echo "$name rc=$?"; }
RGCN_HOT_KB=32 b hot32
RGCN_HOT_KB=16 b hot16
RGCN_HOT_KB=0 b hot0
python3 - <<PY
import json
for n in ("hot32", "hot16", "hot0"):
    try:
        r = json.loads([l for l in open("$out/${tag}_%s.json" % n) if l.startswith("{")][-1])
        print(n, round(r["ms_per_step"], 4), [(k["kernel"], round(k["avg_us"], 1)) for k in r["gather_kernels"]])
    except Exception as exc:
        print(n, "unreadable:", exc)
PY

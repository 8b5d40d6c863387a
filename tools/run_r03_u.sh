#!/bin/bash
# end-of-round rehearsal: what the driver runs (GPU tier, smoke, the default bench line), timed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03u}
t0=$(date +%s)
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
t1=$(date +%s)
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1 || { tail -20 $out/${tag}_smoke.log; exit 1; }
tail -2 $out/${tag}_smoke.log
t2=$(date +%s)
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -20 $out/${tag}_bench.err; exit 1; }
t3=$(date +%s)
echo "pytest $((t1 - t0)) s, smoke $((t2 - t1)) s, bench $((t3 - t2)) s"
python3 - <<PY
import json
r = json.load(open("$out/${tag}_bench.json"))
print(r["ms_per_step"], r["value"], r["roofline"]["frac"], r["dominant_kernel"]["kernel"] if "kernel" in r["dominant_kernel"] else r["dominant_kernel"])
print("fp32", r.get("fp32_mfma_ms_per_step"), "c4", r["secondary"]["c4_1gpu"].get("ms_per_step"), "cpu", r["cpu_baseline"]["value"])
PY

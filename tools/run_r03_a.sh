#!/bin/bash
# round 3, first GPU pass: the whole -m gpu tier, the headline bench line (with its secondary legs), the N > 1 legs
# rehearsed with two gloo ranks on the one GPU (C2 and C4), the host-side profile of an eager encoder step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03a}
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -5 $out/${tag}_pytest.log
timeout -k 10 400 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
tail -c 600 $out/${tag}_bench.err
export RGCN_BENCH_BACKEND=gloo
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 3 > $out/${tag}_gloo2_c2.json 2> $out/${tag}_gloo2_c2.err; echo "gloo2 c2 rc=$?"
tail -c 400 $out/${tag}_gloo2_c2.err
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 2 --steps 5 --warmup 2 --workload c4 > $out/${tag}_gloo2_c4.json 2> $out/${tag}_gloo2_c4.err; echo "gloo2 c4 rc=$?"
tail -c 400 $out/${tag}_gloo2_c4.err
unset RGCN_BENCH_BACKEND
python3 tools/host_profile.py > $out/${tag}_host_profile.txt 2>&1; tail -5 $out/${tag}_host_profile.txt
python3 - <<PY
import json
for n in ("bench", "gloo2_c2", "gloo2_c4"):
    try:
        r = json.load(open("$out/${tag}_%s.json" % n))
    except Exception as exc:
        print(n, "unreadable:", exc); continue
    print(n, r["ms_per_step"], r["value"], r["config"]["launch"], r.get("roofline", {}).get("frac"), r.get("roofline", {}).get("bound"))
    print("  dominant:", r.get("dominant_kernel"))
    print("  fp32:", r.get("fp32_mfma_ms_per_step"), " secondary:", json.dumps(r.get("secondary"))[:900])
PY

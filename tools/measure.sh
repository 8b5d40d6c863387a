#!/bin/bash
# One measurement pass on the GPU box: headline bench line (with its fp32 / C4 secondary legs), rocprofv3 kernel stats of
# the same command without the secondary legs (split and fp32 arithmetic, C3, C4 on one GPU, E = 1.68 M), PMC passes
# (separate runs, --kernel-trace only beside --pmc), the N > 1 legs rehearsed with two gloo ranks on the one GPU.
# usage: bash tools/measure.sh <tag> [1|2|3]   -> gpurun_out/<tag>_*   (part 1: bench lines + kernel stats; part 2: PMC
#        passes; part 3: gloo rehearsals, host profile, stamps, epoch, eval - each part fits one gpurun call of <= 20 minutes)
tag=${1:-r04}
part=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
if [ "$part" = 1 ]; then
python3 bench.py > $out/${tag}_c2_bench.json 2> $out/${tag}_c2_bench.err
stats() {  # name, extra bench args ...: the line itself first (no profiler attached), then the same command under rocprofv3
  local name=$1; shift
  [ "$name" = c2 ] || python3 bench.py --no-cpu-baseline --no-secondary "$@" > $out/${tag}_${name}_bench.json 2> $out/${tag}_${name}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$name -o p -- python3 bench.py --no-cpu-baseline --no-secondary "$@" \
      > $out/${tag}_${name}_prof_bench.json 2> $out/${tag}_${name}_prof.err
  cp "$(find $out/prof_${tag}_$name -name '*kernel_stats.csv' | head -1)" $out/${tag}_${name}_kernel_stats.csv
}
stats c2
RGCN_GEMM_PRECISION=fp32 stats c2_fp32
stats c2_fp16 --fp16-gather
stats c3 --workload c3
stats c4_1gpu --workload c4 --steps 10 --warmup 3
RGCN_TRAIN_FUSED=0 stats c4_1gpu_unfused --workload c4 --steps 10 --warmup 3
stats c2_e1677772 --edges 1677772
fi
if [ "$part" = 2 ]; then
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  d=$out/pmc_${tag}/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 \
      > /dev/null 2> $out/${tag}_pmc.err
done
python3 tools/pmc_summary.py $out/pmc_${tag} $out/${tag}_pmc_counters.json "C2 bench (split precision), rocprofv3 --pmc, one counter set per pass" > $out/${tag}_pmc_summary.txt
fi
if [ "$part" = 3 ]; then
# (every step says so on stdout: a command that is silent for 7 minutes is taken to be hung - and no pipes into grep / tail)
export RGCN_BENCH_BACKEND=gloo
echo "gloo rehearsal, 2 ranks, C2 (+ its secondary C4 leg)"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 10 --warmup 3 > $out/${tag}_gloo2_c2_bench.json 2> $out/${tag}_gloo2_c2.err
echo "gloo rehearsal, 2 ranks, C4"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 2 --steps 5 --warmup 2 --workload c4 --no-secondary > $out/${tag}_gloo2_c4_bench.json 2> $out/${tag}_gloo2_c4.err
unset RGCN_BENCH_BACKEND
echo "host profile"; python3 tools/host_profile.py 12 > $out/${tag}_host_profile.txt 2>&1
echo "stamps"; tools/gemm_stamps > $out/${tag}_stamps.txt 2>&1
echo "epoch"; python3 tools/epoch_time.py > $out/${tag}_epoch.txt 2>&1
python3 tools/epoch_time.py --no_hip_graph >> $out/${tag}_epoch.txt 2>&1
echo "eval"; python3 tools/eval_time.py > $out/${tag}_eval.txt 2>&1
fi
python3 - <<PY
import json
import os, sys
if not os.path.exists("$out/${tag}_c2_bench.json"): sys.exit(0)
r = json.load(open("$out/${tag}_c2_bench.json"))
print("C2", r["ms_per_step"], r["value"], r["config"]["launch"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "bound", "avg_us", "achieved", "frac", "frac_compulsory")})
print("dominant", r["dominant_kernel"])
print("mfma", {k: r["roofline_mfma"][k] for k in ("call", "arithmetic", "avg_us", "achieved", "frac", "executed_tflops", "sum_transform_us_per_step")})
print("fp32", r.get("fp32_mfma_ms_per_step"), "c4_1gpu", {k: r["secondary"]["c4_1gpu"].get(k) for k in ("ms_per_step", "value", "gather_vs_hbm_peak")})
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["gpu_over_cpu"])
for n in ("c2_fp32", "c2_fp16", "c3", "c4_1gpu", "c4_1gpu_unfused", "c2_e1677772", "gloo2_c2", "gloo2_c4"):
    try:
        q = json.loads([l for l in open(f"$out/${tag}_{n}_bench.json").read().splitlines() if l.startswith("{")][-1])
        print(n, q["ms_per_step"], q["value"], q.get("roofline", {}).get("frac"), q.get("roofline", {}).get("kernel"))
    except Exception as exc:
        print(n, "unreadable", exc)
PY

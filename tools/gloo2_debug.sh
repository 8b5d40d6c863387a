#!/bin/bash
# two gloo ranks of bench.py on the one GPU, each under faulthandler: a rank that hangs is aborted after $1 seconds and
# leaves the Python stack of every thread in gpurun_out/gloo2_debug_rank*.err
cd "$GRAFT_REPO_ROOT" || exit 1
export RGCN_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT ${1:-150} python3 -X faulthandler bench.py --gpus 2 --steps 5 --warmup 2 ${@:2} \
      > gpurun_out/gloo2_debug_rank$r.out 2> gpurun_out/gloo2_debug_rank$r.err &
done
wait
echo done

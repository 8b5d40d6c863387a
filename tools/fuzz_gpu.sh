#!/bin/bash
# GPU box: the four fuzz scripts on fresh seeds against the library as built (progress lines, no pipes); summaries -> gpurun_out/<tag>_fuzz.txt
tag=${1:-r04}
mkdir -p gpurun_out
out=gpurun_out/${tag}_fuzz.txt
: > $out
run() {  # name, seconds, command ...
  local name=$1 limit=$2; shift 2
  echo "$name"
  timeout -k 10 $limit "$@" > gpurun_out/${tag}_fuzz_$name.log 2>&1
  echo "== $name (rc $?): $*" >> $out
  grep -E "^FAIL|cases in|worst|all " gpurun_out/${tag}_fuzz_$name.log >> $out || true
}
run encoder_a 500 python tools/fuzz_encoder.py 160 9001
run encoder_b 500 python tools/fuzz_encoder.py 160 9002
run dropout 300 python tools/fuzz_dropout.py 40 9003
run head 300 python tools/fuzz_head.py 150 9004
run dist 400 python tools/fuzz_dist.py 100 9005
cat $out

#!/bin/bash
# A/B of the head's backward scatter: chunked two-pass (default) against the one-pass form (RGCN_SCATTER=single)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03r}
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_train.py -m gpu -q -k "distmult or train or adam or bce or link" > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
export RGCN_EPOCH_STEPS=300
for mode in chunked single; do
  if [ $mode = single ]; then export RGCN_SCATTER=single; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$mode -o p -- python3 tools/epoch_time.py > $out/${tag}_${mode}_prof.log 2>&1
  cp "$(find $out/prof_${tag}_$mode -name '*kernel_stats.csv' | head -1)" $out/${tag}_${mode}_kernel_stats.csv
  echo "== $mode"; grep "epoch" $out/${tag}_${mode}_prof.log | grep -v amdgpu
  grep -E "k_scatter|k_segment|k_adam|k_sumsq|FillFunctor" $out/${tag}_${mode}_kernel_stats.csv | awk -F, '{printf "%s %s calls %.2f us\n", substr($1,1,60), $2, $4/1000}'
done

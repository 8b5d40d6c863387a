#!/bin/bash
# eager epoch with and without the native step, same box; full GPU suite first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03m}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -30 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
for i in 1 2; do
  echo "== eager, native step" >> $out/${tag}_epoch.txt
  python3 tools/epoch_time.py --no_hip_graph 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
  echo "== eager, wrappers" >> $out/${tag}_epoch.txt
  RGCN_NATIVE_STEP=0 python3 tools/epoch_time.py --no_hip_graph 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
done
echo "== graph replay" >> $out/${tag}_epoch.txt
python3 tools/epoch_time.py 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
cat $out/${tag}_epoch.txt

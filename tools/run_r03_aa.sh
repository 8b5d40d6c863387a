#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03aa}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
python3 tools/epoch_time.py 2>&1 | grep -v amdgpu > $out/${tag}_epoch.txt
python3 tools/epoch_time.py --no_hip_graph 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
cat $out/${tag}_epoch.txt

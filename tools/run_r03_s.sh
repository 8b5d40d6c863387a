#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03s}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1 || { tail -40 $out/${tag}_pytest.log; exit 1; }
tail -2 $out/${tag}_pytest.log
python3 tools/epoch_time.py 2>&1 | grep -v amdgpu > $out/${tag}_epoch.txt
python3 tools/epoch_time.py --no_hip_graph 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
python3 tools/epoch_time.py 2>&1 | grep -v amdgpu >> $out/${tag}_epoch.txt
cat $out/${tag}_epoch.txt
export RGCN_EPOCH_STEPS=300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_train -o p -- python3 tools/epoch_time.py > $out/${tag}_train_prof.log 2>&1
cp "$(find $out/prof_${tag}_train -name '*kernel_stats.csv' | head -1)" $out/${tag}_train_kernel_stats.csv

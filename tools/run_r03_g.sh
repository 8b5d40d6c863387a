#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out
tag=${1:-r03g}
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/${tag}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest.log
tail -8 $out/${tag}_pytest.log
python3 tools/epoch_time.py > $out/${tag}_epoch.txt 2>&1; tail -3 $out/${tag}_epoch.txt
python3 tools/epoch_time.py --no_hip_graph > $out/${tag}_epoch_eager.txt 2>&1; tail -3 $out/${tag}_epoch_eager.txt
RGCN_NATIVE_STEP=0 python3 tools/epoch_time.py --no_hip_graph > $out/${tag}_epoch_eager_wrappers.txt 2>&1; tail -3 $out/${tag}_epoch_eager_wrappers.txt
python3 tools/eval_time.py > $out/${tag}_eval.txt 2>&1; tail -5 $out/${tag}_eval.txt
python3 tools/host_profile.py 14 > $out/${tag}_host_profile.txt 2>&1; head -20 $out/${tag}_host_profile.txt

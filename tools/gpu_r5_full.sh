#!/bin/bash
# the whole -m gpu suite + smoke on the current tree, then the profile refresh
tag=${1:-r5full}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -25 $out/pytest.log | cut -c1-300
bash tools/refresh_profiles.sh $tag > $out/refresh.log 2>&1; echo "refresh rc=$?"
ls $out | wc -l

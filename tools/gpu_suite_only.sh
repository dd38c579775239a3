#!/bin/bash
# the -m gpu suite + smoke on the current tree (the log goes to profiles/ as rNN_gpu_pytest.log)
tag=${1:-suite}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log | cut -c1-200

#!/bin/bash
# One GPU-box visit of the build loop: the -m gpu suite with durations, smoke, the driver's bench invocation and the C5 loopback bench.
# usage: bash tools/gpu_round.sh <tag> [pytest args...]
tag=${1:-r03}; shift
out=gpurun_out/$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
timeout 3000 python -m pytest tests -m gpu -q --durations=30 "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/smoke.log; tail -60 $out/pytest.log

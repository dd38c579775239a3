#!/bin/bash
# The 8-GPU day in one command: see tools/gpu_scale.py (this wrapper only sets the environment a gpurun box needs).
#   bash tools/gpu_scale.sh [outdir] [steps]        bash tools/gpu_scale.sh --dry
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$1" = "--dry" ]; then exec python tools/gpu_scale.py --dry; fi
exec python tools/gpu_scale.py --out ${1:-gpurun_out/scale} --steps ${2:-200}

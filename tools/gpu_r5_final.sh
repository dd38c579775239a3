#!/bin/bash
# round 5, final tree: suite + smoke + profile refresh (gpu_r5_full.sh), then the two-queue soak and the determinism repeats
tag=${1:-r5final2}
bash tools/gpu_r5_full.sh $tag
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python tools/soak_overlap.py 320 60 > $out/soak.log 2>&1; tail -2 $out/soak.log
timeout 600 python tools/determinism_stress.py > $out/determinism.log 2>&1; tail -2 $out/determinism.log

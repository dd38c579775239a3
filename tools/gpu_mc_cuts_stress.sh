#!/bin/bash
# phase cuts of the one-pass marching cubes with every voxel of the 128^3 grid meshed (build the cut libraries as tools/gpu_mc_cuts.sh says:
# from commit 3efda2c, the last one that carries the DIF_MC_CUT blocks)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1 2 3 4 full; do
  if [ $c = full ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_mccut$c.so; fi
  timeout 600 python tools/stress_full_occupancy.py --reps 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cut $c', [r['mc_count_ms'] for r in d['runs']])"
done

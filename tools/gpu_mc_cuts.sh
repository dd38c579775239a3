#!/bin/bash
# phase cuts of the one-pass marching cubes: event-timed launch per cut on the stream and the K = 20 run (stress: tools/stress_full_occupancy.py with DIF_LIB set the same way).
# The cuts (#if DIF_MC_CUT blocks in mc_onepass_body) lived in the tree up to commit 3efda2c; the kernel has since been split into
# mc_onepass_direct / mc_onepass_ring without them: check that commit out to rebuild the cut libraries.
# Build the cut libraries HERE first (they travel with the snapshot, ab_old/ is git-ignored):
#   for c in 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DNDEBUG -DDIF_MC_CUT=$c \
#       -DDIF_BUILD_ID='"cut"' di_fusion_amd/csrc/difusion.hip -o ab_old/libdif_mccut$c.so; done
# (tools/gpu_mc_cuts_stress.sh: the same cuts on the full-occupancy stress)
# (a cut build's results are garbage by design: 1 = launch + neighbour look-ups, 2 = + blended corners, 3 = + cells, 4 = + look-back, no emit)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1 2 3 4 full; do
  if [ $c = full ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_mccut$c.so; fi
  for args in "" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu-baseline --no-secondary --d2h none $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cut $c', '$args', d['roofline']['other_ms_per_frame'], d['ms_per_step'])"
  done
done

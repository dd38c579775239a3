#!/bin/bash
# phase cuts of the one-pass marching cubes (-DDIF_MC_CUT=1..4 builds in ab_old/): event-timed launch per cut on the stream and the K = 20 run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1 2 3 4 full; do
  if [ $c = full ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_mccut$c.so; fi
  for args in "" "--steps 20 --warmup 5"; do
    timeout 200 python bench.py --no-cpu-baseline --no-secondary --d2h none $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cut $c', '$args', d['roofline']['other_ms_per_frame'], d['ms_per_step'])"
  done
done

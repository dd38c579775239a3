#!/bin/bash
# A/B of marching-cubes builds on the full-occupancy stress: ab_old/libdif_<name>.so against the in-tree library ("main")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@" main; do
  if [ $v = main ]; then unset DIF_LIB; else export DIF_LIB=$GRAFT_REPO_ROOT/ab_old/libdif_$v.so; fi
  timeout 600 python tools/stress_full_occupancy.py --reps 3 $STRESS_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', [(r['mc_count_ms'], r['mc_algorithmic_GBps'], r['T']) for r in d['runs']])"
done

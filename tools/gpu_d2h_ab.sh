#!/bin/bash
# the per-frame triangle export: carried by kernels (new), by the copy engine behind an event (dma), or not at all (none)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in new dma none; do
  for args in "" "--steps 20 --warmup 5"; do
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --d2h $m $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', '$args', d['value'], d['ms_per_step'])"
  done
done
done

#!/bin/bash
out=gpurun_out/r5final2_rep; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --streams-per-gpu 2 2>/dev/null | tail -1 > $out/bench_s2_$i.json
  DIF_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $out/bench_rccl_1rank_$i.json
  DIF_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary --rccl-before-clock 1 2>/dev/null | tail -1 > $out/bench_rccl_1rank_before_clock_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5final2_rep/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY

#!/bin/bash
out=gpurun_out/fence1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_long.py tests/test_gpu_track.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $out/b$i.json; timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | tail -1 > $out/k$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/fence1/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
timeout 300 python tools/bench_query.py 2>/dev/null | tail -1

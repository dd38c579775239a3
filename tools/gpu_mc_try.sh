#!/bin/bash
# after a marching-cubes change: its parity tests, the full-occupancy stress (every voxel of the 128^3 grid meshed), the stream bench lines
tag=${1:-mc}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_mc_exhaustive.py tests/test_gpu_mesh_anchor.py tests/test_gpu_map.py tests/test_gpu_stream.py "tests/test_gpu_fuzz.py::test_fuzz_marching_cubes" -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/stress_full_occupancy.py --reps 3 2>/dev/null | tail -1 > $out/stress.json
python - $out/stress.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for r in d["runs"]: print("stress", r["mc_count_ms"], r["mc_emit_ms"], r["mc_algorithmic_GBps"], r["T"])
PY
for args in "" "--steps 20 --warmup 5" "--streams-per-gpu 8"; do for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-secondary --d2h none $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', '$args', d['value'], d['ms_per_step'], d['roofline'].get('other_ms_per_frame'))"
done; done

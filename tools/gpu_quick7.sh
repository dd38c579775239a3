#!/bin/bash
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_map.py tests/test_gpu_pipes.py tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -4
for v in 0 1; do echo "== DIF_VD_DUAL=$v"; DIF_VD_DUAL=$v python tools/sweep_decode.py --sizes 1,512,768,1024,2048,4096 --reps 7 2>&1 | grep decode_lattice | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['n'], d['decode_lattice_us'], d['decode_points_us'], d['mc_count_us'])"; done
for v in 0 1 0 1; do DIF_VD_DUAL=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $out/b.json
python - $out/b.json "dual=$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], {k:v["ms_per_launch"] for k,v in r["per_kernel"].items()})
PY
done
for v in 0 1; do DIF_VD_DUAL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $out/b.json
python - $out/b.json "k20 dual=$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], {k:v["ms_per_launch"] for k,v in r["per_kernel"].items()})
PY
done

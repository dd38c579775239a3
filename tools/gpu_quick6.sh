#!/bin/bash
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parallel.py tests/test_gpu_map.py -m gpu -q -x 2>&1 | tail -6
for args in "" "--steps 20 --warmup 5" "" "--steps 20 --warmup 5" "--graph 1" "--mode tiled --loopback 8 --steps 100"; do timeout 300 python bench.py $args --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $out/b.json
python - $out/b.json "$args" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(repr(sys.argv[2]), d["value"], d["ms_per_step"], r["other_ms_per_frame"])
PY
done
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 | grep -i "onepass\|finish\|sum of\|unproject"
rm -rf $out/trace

#!/bin/bash
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_multiproc.py tests/test_gpu_stream.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -25
DIF_LIB=tools/libdifusion_trace.so timeout 300 python tools/trace_decode.py --frames 160 > $out/trace160.json 2> $out/trace160.err; cat $out/trace160.json; tail -2 $out/trace160.err
for args in "--mode tiled --loopback 8" "--mode tiled --loopback 8 --halo full" "--mode tiled --loopback 2" "--mode tiled"; do
  n=$(echo $args | tr -d ' -'); timeout 300 python bench.py $args --no-cpu-baseline --steps 100 > $out/$n.json 2> $out/$n.err
  python - $out/$n.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("halo_exchange"), (d.get("roofline") or {}).get("other_ms_per_frame"))
except Exception as e: print("ERR",e)
PY
  tail -2 $out/$n.err
done

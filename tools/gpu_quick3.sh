#!/bin/bash
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_map.py tests/test_gpu_pipes.py tests/test_gpu_stream.py tests/test_gpu_parallel.py -m gpu -q -x 2>&1 | tail -8
DIF_LIB=tools/libdifusion_trace.so timeout 300 python tools/trace_decode.py --frames 160 > $out/trace160.json 2> $out/trace160.err; cat $out/trace160.json; tail -2 $out/trace160.err
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $out/bench_n1.json 2> $out/bench_n1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_k20.json 2> $out/bench_k20.err
for f in bench_n1 bench_k20; do python - $out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["ms_per_launch"],v["frac"]) for k,v in (r.get("per_kernel") or {}).items()}, r.get("other_ms_per_frame"))
except Exception as e: print("ERR",e)
PY
done
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python bench.py --no-cpu-baseline --no-secondary > $out/trace.log 2>&1
python tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 200 > $out/kernel_stats.md 2>&1
rm -rf $out/trace; cat $out/kernel_stats.md

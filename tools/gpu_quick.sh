#!/bin/bash
# quick look: driver bench line + C5 loopback (delta / full) + plain tiled
tag=${1:-q}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_k20.json 2> $out/bench_k20.err
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $out/bench_n1.json 2> $out/bench_n1.err
timeout 300 python bench.py --mode tiled --no-cpu-baseline --steps 100 > $out/tiled_n1.json 2> $out/tiled_n1.err
timeout 300 python bench.py --mode tiled --loopback 8 --no-cpu-baseline --steps 100 > $out/tiled_lb8_delta.json 2> $out/tiled_lb8_delta.err
timeout 300 python bench.py --mode tiled --loopback 8 --halo full --no-cpu-baseline --steps 100 > $out/tiled_lb8_full.json 2> $out/tiled_lb8_full.err
timeout 300 python bench.py --mode tiled --loopback 2 --no-cpu-baseline --steps 100 > $out/tiled_lb2_delta.json 2> $out/tiled_lb2_delta.err
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace_lb8 -o bench -- python bench.py --mode tiled --loopback 8 --no-cpu-baseline --steps 100 > $out/trace_lb8.log 2>&1
python tools/rocpd_stats.py $(find $out/trace_lb8 -name "*.db" | head -1) --after-nth k_prune_mark 14 --frames 100 > $out/kernel_stats_lb8.md 2>&1
rm -rf $out/trace_lb8
for f in $out/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["config"].get("halo_exchange"), (d.get("roofline") or {}).get("other_ms_per_frame"))
    r=d.get("roofline") or {}
    print({k:(v["ms_per_launch"],v["frac"]) for k,v in (r.get("per_kernel") or {}).items()})
except Exception as e: print("ERR",e)
PY
done
tail -3 $out/*.err | tail -30
cat $out/kernel_stats_lb8.md

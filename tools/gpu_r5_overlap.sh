#!/bin/bash
# round 5: two queues per stream (frame i+1's front end beside frame i's extract), host depth, SDMA export, C5 culling: parity, A/B, timeline
tag=${1:-r5e}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parallel.py -m gpu -q -x > $out/pytest_stream.log 2>&1; echo "stream+parallel tests rc=$?"; tail -6 $out/pytest_stream.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_long.py -m gpu -q -x > $out/pytest_long.log 2>&1; echo "long rc=$?"; tail -3 $out/pytest_long.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(f"{sys.argv[2]:26s} {d['value']:9.1f} frames/s {d['ms_per_step']} ms engine={str(c.get('d2h_engine'))[:6]} two_queues={c.get('two_queues',{}).get('on')} depth={c.get('host_pipeline_depth')} "
          + str({k: (round(v['ms_per_launch'] * 1e3, 1), v['frac']) for k, v in r['per_kernel'].items()}))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
  run ov1_d2_200_$rep $B
  run ov1_d1_200_$rep $B --host-depth 1
  run ov0_d2_200_$rep $B --overlap 0
  run ov0_d1_200_$rep $B --overlap 0 --host-depth 1
  run ov1_d2_k20_$rep $B --steps 20 --warmup 5
  run ov0_d1_k20_$rep $B --overlap 0 --host-depth 1 --steps 20 --warmup 5
done
run ov1_d2_none $B --d2h none
run tiled_lb8 python bench.py --mode tiled --loopback 8 --no-cpu-baseline --steps 100
run tiled_n1 python bench.py --mode tiled --no-cpu-baseline --steps 100
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- $B > $out/trace_bench.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $db --after-nth k_prune_mark 160 --frames 50 > $out/kernel_stats_steady.md 2>&1
python tools/rocpd_stats.py $db --timeline k_prune_mark 150 > $out/timeline_overlap.txt 2>&1
rm -rf $out/trace
head -24 $out/kernel_stats_steady.md; head -48 $out/timeline_overlap.txt

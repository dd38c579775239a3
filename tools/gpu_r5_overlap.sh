#!/bin/bash
# round 5: two queues per stream (frame i+1's front end beside frame i's extract) and the SDMA export: parity, A/B, RCCL both sides of the clock, timeline
tag=${1:-r5d}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_stream.py -m gpu -q -x > $out/pytest_stream.log 2>&1; echo "stream tests rc=$?"; tail -6 $out/pytest_stream.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_long.py -m gpu -q -s -x > $out/pytest_long.log 2>&1; echo "long rc=$?"; tail -4 $out/pytest_long.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print(f"{sys.argv[2]:26s} {d['value']:9.1f} frames/s {d['ms_per_step']} ms engine={str(c.get('d2h_engine'))[:14]} two_queues={c.get('two_queues',{}).get('on')} rccl={d.get('rccl_ranks')}/{c.get('rccl_before_clock')} "
          + str({k: (round(v['ms_per_launch'] * 1e3, 1), v['frac']) for k, v in r['per_kernel'].items()}))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for rep in 1 2; do
  run ov1_200_$rep $B
  run ov0_200_$rep $B --overlap 0
  run ov1_k20_$rep $B --steps 20 --warmup 5
  run ov0_k20_$rep $B --overlap 0 --steps 20 --warmup 5
done
run ov1_none $B --d2h none
run ov0_none $B --overlap 0 --d2h none
run s4_dma_ov1 $B --streams-per-gpu 4 --d2h dma
run s4_new $B --streams-per-gpu 4 --d2h new
run s8_dma_ov1 $B --streams-per-gpu 8 --d2h dma
run s8_new $B --streams-per-gpu 8 --d2h new
DIF_FORCE_DIST=1 run rccl_behind $B --rccl-before-clock 0
DIF_FORCE_DIST=1 run rccl_before $B --rccl-before-clock 1
DIF_FORCE_DIST=1 run rccl_before_ov0 $B --rccl-before-clock 1 --overlap 0
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- $B > $out/trace_bench.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $db --after-nth k_prune_mark 160 --frames 50 > $out/kernel_stats_steady.md 2>&1
python tools/rocpd_stats.py $db --timeline k_prune_mark 150 > $out/timeline_overlap.txt 2>&1
rm -rf $out/trace
head -24 $out/kernel_stats_steady.md; head -48 $out/timeline_overlap.txt

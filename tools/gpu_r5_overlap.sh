#!/bin/bash
# round 5: two queues per stream (frame i+1's front end beside frame i's extract) and the SDMA export: parity, A/B, timeline
tag=${1:-r5b}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_stream.py -m gpu -q -x -k "two_queue or copy_engine or growth_after or rejects" > $out/pytest_ov.log 2>&1; echo "overlap tests rc=$?"; tail -15 $out/pytest_ov.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_long.py -m gpu -q -s -x > $out/pytest_long.log 2>&1; echo "long rc=$?"; tail -12 $out/pytest_long.log | cut -c1-400
B="python bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  for ov in 1 0; do
    timeout 300 $B --overlap $ov > $out/bench200_ov${ov}_$rep.json 2> $out/bench200_ov${ov}_$rep.err
    timeout 300 $B --overlap $ov --steps 20 --warmup 5 > $out/bench20_ov${ov}_$rep.json 2> $out/bench20_ov${ov}_$rep.err
  done
done
timeout 300 $B --overlap 1 --d2h none > $out/bench200_ov1_none.json 2> $out/bench200_ov1_none.err
timeout 300 $B --overlap 0 --d2h none > $out/bench200_ov0_none.json 2> $out/bench200_ov0_none.err
python - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1] + "/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], c.get("d2h_engine"), c.get("two_queues", {}).get("on"), c.get("two_queues", {}).get("queues_independent"),
              {k: (round(v["ms_per_launch"] * 1e3, 1), v["frac"]) for k, v in r["per_kernel"].items()})
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-600:])
PY
timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- $B --overlap 1 > $out/trace_bench.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
python tools/rocpd_stats.py $db --after-nth k_prune_mark 160 --frames 50 > $out/kernel_stats_steady.md 2>&1
python tools/rocpd_stats.py $db --timeline k_prune_mark 150 > $out/timeline_overlap.txt 2>&1
rm -rf $out/trace
head -30 $out/kernel_stats_steady.md; head -45 $out/timeline_overlap.txt

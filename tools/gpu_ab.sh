#!/bin/bash
# A/B: the committed tree (ab_old/, its own library) against the working tree, same box, steady-state frames only
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in old new old new; do
  d=$GRAFT_REPO_ROOT; [ $v = old ] && d=$GRAFT_REPO_ROOT/ab_old
  (cd $d && timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1) > $out/bench_$v.json
  python - $out/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], d["value"], {k:x["avg_launch_ms"] for k,x in r["by_phase"]["second_half_of_timed_frames"].items()}, r["other_ms_per_frame"])
PY
done
for v in old new; do
  d=$GRAFT_REPO_ROOT; [ $v = old ] && d=$GRAFT_REPO_ROOT/ab_old
  (cd $d && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace_$v -o bench -- python bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1)
  python tools/rocpd_stats.py $(find $out/trace_$v -name "*.db" | head -1) --after-nth k_prune_mark 160 --frames 50 > $out/steady_$v.md 2>&1
  rm -rf $out/trace_$v; echo "== $v"; cat $out/steady_$v.md
done

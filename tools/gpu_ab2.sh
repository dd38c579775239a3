#!/bin/bash
# A/B of two builds on one box: the tree's library against ab_old/<name>.so (DIF_LIB), interleaved runs.  usage: bash tools/gpu_ab2.sh <tag> <other.so> "<bench args>" [reps]
tag=$1; other=$2; args=$3; reps=${4:-3}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq 1 $reps); do
  for which in new old; do
    if [ $which = old ]; then export DIF_LIB=$GRAFT_REPO_ROOT/$other; else unset DIF_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-secondary $args 2>/dev/null | tail -1 > $out/${which}_$i.json
    python - $out/${which}_$i.json $which <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], {k:round(v["ms_per_launch"]*1e3,1) for k,v in r["per_kernel"].items()}, r["other_ms_per_frame"])
PY
  done
done

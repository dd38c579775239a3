#!/bin/bash
# A/B on one box: the frame's decoder as ONE persistent launch (default) against the two launches of rounds 2-5 (DIF_DECODE_LAUNCHES=2).
# usage (via gpurun): bash tools/gpu_ab_decode.sh <tag> [reps]
tag=${1:-ab_decode}; reps=${2:-3}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in $(seq 1 $reps); do
  for L in 1 2; do
    DIF_DECODE_LAUNCHES=$L timeout 300 python bench.py --no-cpu-baseline --no-secondary > $out/b200_L${L}_$r.json 2> $out/b200_L${L}_$r.err
    DIF_DECODE_LAUNCHES=$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $out/k20_L${L}_$r.json 2> $out/k20_L${L}_$r.err
    DIF_DECODE_LAUNCHES=$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --overlap 0 > $out/b200q1_L${L}_$r.json 2> $out/b200q1_L${L}_$r.err
  done
done
python - $out <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for kind in ("b200", "k20", "b200q1"):
    for L in (1, 2):
        vals = []
        for f in sorted(glob.glob(f"{out}/{kind}_L{L}_*.json")):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((d["value"], d["ms_per_step"], d["roofline"]["per_kernel"] if d.get("roofline") else None))
            except Exception as e:
                vals.append(("ERR", repr(e)[:60], None))
        print(kind, f"launches={L}", [(v[0], v[1]) for v in vals])
        if vals and vals[-1][2]:
            print("    per_kernel:", json.dumps(vals[-1][2])[:600])
PY

#!/bin/bash
# A/B on one box of the two-queue arrangement: where the overlapped frame's encoder starts (DIF_OV_ENC_WAIT) and which shape it has (DIF_OV_ENC_SLIM).
# usage (via gpurun): bash tools/gpu_ab_overlap.sh <tag> [reps]
tag=${1:-ab_overlap}; reps=${2:-2}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for r in $(seq 1 $reps); do
  for W in 0 1; do for S in 0 1; do
    DIF_OV_ENC_WAIT=$W DIF_OV_ENC_SLIM=$S timeout 300 python bench.py --no-cpu-baseline --no-secondary > $out/b200_w${W}s${S}_$r.json 2> $out/b200_w${W}s${S}_$r.err
    DIF_OV_ENC_WAIT=$W DIF_OV_ENC_SLIM=$S timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $out/k20_w${W}s${S}_$r.json 2> $out/k20_w${W}s${S}_$r.err
  done; done
done
python - $out <<'PY'
import json, sys, glob
out = sys.argv[1]
for kind in ("b200", "k20"):
    for W in (0, 1):
        for S in (0, 1):
            vals = []
            for f in sorted(glob.glob(f"{out}/{kind}_w{W}s{S}_*.json")):
                try:
                    d = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((d["value"], d["ms_per_step"], {k: round(v["ms_per_launch"] * 1e3, 1) for k, v in d["roofline"]["per_kernel"].items()}))
                except Exception as e:
                    vals.append(("ERR", repr(e)[:80]))
            print(kind, f"wait={W} slim={S}", vals)
PY

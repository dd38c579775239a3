#!/bin/bash
# marching cubes with every voxel of the 128^3 grid meshed (tools/stress_full_occupancy.py): HBM traffic and wait cycles of the one-pass kernel,
# one rocprofv3 --pmc pass per counter set (never combined with a trace)
tag=${1:-mcpmc}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python tools/stress_full_occupancy.py --reps 2 > $out/p$i.log 2>&1
done
python - $out $(find $out -name "*counter_collection.csv") <<'PY'
import sys, csv, json, collections, re
out = sys.argv[1]
per = collections.defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if "marching_cubes_onepass" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v[-2:]) / len(v[-2:]) for k, v in per.items()}      # the two timed repetitions
runs = json.loads([l for l in open(out + "/p1.log") if l.startswith('{"workload"')][-1])["runs"]
K, B, T = runs[-1]["K"], runs[-1]["B"], runs[-1]["T"]
alg = B * 2 * 512 * 4 + T * 56                  # (every triangle is written: --max-triangles above T)
res = {"note": "rocprofv3 --pmc, one pass per counter set, `python tools/stress_full_occupancy.py --reps 2`: k_marching_cubes_onepass<4> over every voxel of the "
               "128^3 grid (ticket mode); averages of the two timed launches.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled as "
               "/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950); algorithmic = B*2*512*4 (the sdf and std cubes once) + T*56 (the triangle rows).",
       "K": K, "B": B, "T": T, "counters": {k: round(v, 1) for k, v in sorted(c.items())},
       "algorithmic_bytes": alg}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    res["hbm_bytes"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    res["traffic_over_algorithmic"] = round(res["hbm_bytes"] / alg, 3)
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    res["wait_any_over_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
res["mc_ms_under_pmc"] = [r["mc_count_ms"] for r in runs]
json.dump(res, open(out + "/pmc_mc_stress.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/p1 $out/p2 $out/p3 $out/p4

#!/usr/bin/env python3
"""Wave-cycle breakdown per kernel from one rocprofv3 --pmc pass of SQ counters (CSV): parked / issue-stalled / issuing shares of
SQ_WAVE_CYCLES (guide: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).  Usage: python tools/pmc_sq.py <csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        k = re.sub(r"^void ", "", k)
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "shares of SQ_WAVE_CYCLES, second half of each kernel's dispatches", "kernels": {}}
    for k, c in per.items():
        if k not in ("k_encode", "k_decode_voxels", "k_decode<false>", "k_marching_cubes<false>", "k_marching_cubes<true>") or "SQ_WAVE_CYCLES" not in c:
            continue
        avg = {n: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for n, v in c.items()}
        wc = avg["SQ_WAVE_CYCLES"]
        out["kernels"][k] = {n: round(v / wc, 4) for n, v in avg.items() if n != "SQ_WAVE_CYCLES"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) into HBM bytes per launch per kernel.
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [--last N]"""
import collections
import csv
import json
import re
import sys


def agg(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        k = re.sub(r"^void ", "", k)
        if r["Counter_Name"] == name:
            d[k].append(float(r["Counter_Value"]))
    return d


def tail(v):
    """the dispatches averaged over: the last N (--last N: the timed frames of a short run) or the second half"""
    if "--last" in sys.argv:
        return v[-int(sys.argv[sys.argv.index("--last") + 1]):]
    return v[len(v) // 2:]


def main():
    f = agg(sys.argv[1], "FETCH_SIZE")
    w = agg(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two SEPARATE passes of `python bench.py --no-cpu-baseline --steps 40 --warmup 4 "
                   "` (C3 workload); KB per launch averaged over the second half of the dispatches (or the last N with --last N: the timed frames of the driver's short run). hbm_bytes_per_launch = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024: FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md section HBM (gfx950 reports half of a wide "
                   "coalesced read); WRITE_SIZE uncalibrated.", "kernels": {}}
    for k in f:
        if k.startswith("k_") or k.startswith("dif::"):
            ft = tail(f[k])
            wt = tail(w.get(k, [0.0]))
            fa, wa = sum(ft) / len(ft), sum(wt) / max(1, len(wt))
            out["kernels"][k] = {"FETCH_SIZE_KB": round(fa, 1), "WRITE_SIZE_KB": round(wa, 1), "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k in ("k_decode_voxels", "k_decode<false>", "k_decode", "k_encode", "k_marching_cubes<true>"):
        if k in out["kernels"]:
            print(k, out["kernels"][k])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 --pmc pass (CSV):
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDs * CUs * 4 SIMDs).
On gfx950 GRBM_GUI_ACTIVE comes back summed over the 8 XCDs, and SQ_VALU_MFMA_BUSY_CYCLES is exactly 64 x the number of
v_mfma_f32_32x32x2_f32 issued (checked against the known MFMA count of the full-occupancy decode).
Usage: python tools/pmc_mfma.py <counter_collection.csv> <out.json> [--cus 256] [--xcds 8] [--last N]"""
import collections
import csv
import json
import re
import sys


def main():
    cus = int(sys.argv[sys.argv.index("--cus") + 1]) if "--cus" in sys.argv else 256
    xcds = int(sys.argv[sys.argv.index("--xcds") + 1]) if "--xcds" in sys.argv else 8
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        k = re.sub(r"^void ", "", k)
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, kernel-trace domain only). mfma_util = "
                   "busy cycles / (GPU-active cycles / XCDs x CUs x 4 SIMDs) (GRBM_GUI_ACTIVE is summed over the XCDs; busy cycles = 64 per "
                   "v_mfma_f32_32x32x2_f32); averaged over the second half of the dispatches of each kernel.", "cus": cus, "xcds": xcds,
           "kernels": {}}
    for k, c in per.items():
        if not (k.startswith("k_") or k.startswith("dif::")) or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else None      # the timed frames of a short run
        b = c["SQ_VALU_MFMA_BUSY_CYCLES"][-last:] if last else c["SQ_VALU_MFMA_BUSY_CYCLES"][len(c["SQ_VALU_MFMA_BUSY_CYCLES"]) // 2:]
        g = c["GRBM_GUI_ACTIVE"][-last:] if last else c["GRBM_GUI_ACTIVE"][len(c["GRBM_GUI_ACTIVE"]) // 2:]
        busy, act = sum(b) / len(b), sum(g) / len(g)
        if busy > 0:
            out["kernels"][k] = {"mfma_busy_cycles": round(busy), "gpu_active_cycles_per_xcd": round(act / xcds), "mfma_util": round(busy / (act / xcds * cus * 4), 4)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out["kernels"]))


if __name__ == "__main__":
    main()

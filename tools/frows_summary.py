#!/usr/bin/env python3
"""Per-kernel table for the SURVEY.md section-8f kernels from one rocprofv3 kernel trace (rocpd sqlite) and, optionally, the CSVs of separate
--pmc passes (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE) of the same command.
Usage: python tools/frows_summary.py <trace.db> [--mfma m.csv] [--fetch f.csv] [--write w.csv] [--only substr,substr]"""
import collections
import csv
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:64]


def pmc(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if path:
        for r in csv.DictReader(open(path)):
            d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def arg(name):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else None


def main():
    rows = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels order by start").fetchall()
    only = (arg("--only") or "").split(",") if arg("--only") else None
    agg = collections.OrderedDict()
    for n, s, e in rows:
        k = short(n)
        if only and not any(o in k for o in only):
            continue
        a = agg.setdefault(k, [])
        a.append((e - s) / 1e3)
    m, f, w = pmc(arg("--mfma")), pmc(arg("--fetch")), pmc(arg("--write"))
    print("| kernel | calls | avg us | median us | PMC MFMA busy | HBM MB per launch (2 FETCH + WRITE) |")
    print("|---|---:|---:|---:|---:|---:|")
    for k, ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        ts2 = sorted(ts)
        busy = ""
        if k in m and m[k].get("GRBM_GUI_ACTIVE"):
            b, g = m[k]["SQ_VALU_MFMA_BUSY_CYCLES"], m[k]["GRBM_GUI_ACTIVE"]
            if sum(b) > 0:
                busy = f"{sum(b) / (sum(g) / 8 * 256 * 4):.3f}"
        hbm = ""
        if k in f:
            ff = f[k]["FETCH_SIZE"]; ww = w.get(k, {}).get("WRITE_SIZE", [0.0])
            hbm = f"{(2 * sum(ff) / len(ff) + sum(ww) / max(1, len(ww))) * 1024 / 1e6:.2f}"
        print(f"| `{k}` | {len(ts)} | {sum(ts) / len(ts):.2f} | {ts2[len(ts2) // 2]:.2f} | {busy} | {hbm} |")


if __name__ == "__main__":
    main()

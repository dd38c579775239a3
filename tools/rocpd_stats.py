#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / % — the `--stats` table.
Usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--after-nth KERNEL N [--frames F]] > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<.*)?", name)
    base = m.group(1) if m else name
    if "k_scan_pass" in name:
        f = re.search(r"<(?:.*::)?(\w+Functor)", name)
        base = base + "<" + (f.group(1) if f else "?") + ">"
    elif base.endswith("k_marching_cubes"):
        base += "<emit>" if "true" in name else "<count>"
    elif base.startswith("rocprim"):
        base = "rocprim::" + (re.search(r"(\w+kernel\w*|\w+_kernel)", name).group(1) if re.search(r"(\w+kernel\w*|\w+_kernel)", name) else base)
    return base[:70]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if "--after-nth" in sys.argv:           # e.g. --after-nth k_unproject_transform 5 : drop everything before the 6th frame
        i = sys.argv.index("--after-nth")
        marker, nth = sys.argv[i + 1], int(sys.argv[i + 2])
        starts = [s for n, s, e in rows if marker in n]
        t0 = starts[nth]
        t1 = None
        if "--frames" in sys.argv:          # ... and stop before the (nth + frames)-th one
            nf = int(sys.argv[sys.argv.index("--frames") + 1])
            t1 = starts[nth + nf] if nth + nf < len(starts) else None
        rows = [r for r in rows if r[1] >= t0 and (t1 is None or r[1] < t1)]
        nfr = (len([s for s in starts if s >= t0 and (t1 is None or s < t1)]))
        print(f"(restricted to dispatches from dispatch #{nth} of `{marker}` on: the timed region, {nfr} frames)\n")
    if "--timeline" in sys.argv:            # e.g. --timeline k_prune_mark 150 : every dispatch of two frames, relative times in us
        i = sys.argv.index("--timeline")
        marker, nth = sys.argv[i + 1], int(sys.argv[i + 2])
        all_rows = c.execute("select name, start, end from kernels order by start").fetchall()
        starts = [s for n, s, e in all_rows if marker in n]
        t0, t1 = starts[nth], starts[nth + 2]
        print(f"timeline of two frames from dispatch #{nth} of `{marker}` (us relative to it: start, end, duration)\n")
        for n, s, e in all_rows:
            if t0 - 20000 <= s < t1:
                print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:7.2f}  {short(n)}")
        return
    agg = {}
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"kernel dispatches: {len(rows)}   sum of kernel time: {tot / 1e6:.3f} ms   first-start to last-end: {span / 1e6:.3f} ms\n")
    print("| kernel | calls | total ms | avg us | % of kernel time |")
    print("|---|---:|---:|---:|---:|")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main()

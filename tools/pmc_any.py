#!/usr/bin/env python3
"""Average counter values per kernel (second half of each kernel's dispatches) from rocprofv3 --pmc CSVs.
Usage: python tools/pmc_any.py <csv> [<csv> ...]"""
import collections
import csv
import json
import re
import sys

per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        k = re.sub(r"^void ", "", k)
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(per.items()):
    if not (k.startswith("k_") or k.startswith("dif::")):
        continue
    print(k, json.dumps({n: round(sum(v[len(v) // 2:]) / len(v[len(v) // 2:])) for n, v in sorted(c.items())}))

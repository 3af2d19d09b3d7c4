#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name.
usage: parse_pmc.py <dir-or-csv> [name-filter: substring, or a regular expression after "re:"]"""
import csv, glob, os, re, sys
from collections import defaultdict
csv.field_size_limit(1 << 30)
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mfx"
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: [0.0, 0])
for f in files:
    for r in csv.DictReader(open(f)):
        if (not re.search(flt[3:], r["Kernel_Name"])) if flt.startswith("re:") else (flt not in r["Kernel_Name"]):
            continue
        name = r["Kernel_Name"].split("(mfx")[0].replace("void ", "").replace("mfx::(anonymous namespace)::", "")
        key = (name, r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
for (name, ctr), (s, n) in sorted(acc.items()):
    print(f"{name:32s} {ctr:16s} n={n:5d} mean={s / n:.6g}")

#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs: one line per kernel dispatch (in dispatch order) with every collected counter.
usage: pmc_per_launch.py <dir> [name-filter]"""
import csv, glob, os, sys
from collections import OrderedDict
csv.field_size_limit(1 << 30)
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mfx"
rows = OrderedDict()
for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (did, name), c in sorted(rows.items()):
    print(did, name, " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))

#!/usr/bin/env python3
"""(r4) Reads a rocprofv3 --kernel-trace CSV of a sharded scatter solve and reports how much of the column-side exchange
(k_scatter_combine, the RCCL kernel if any, k_finalize of the column side) ran UNDER a column pass (k_scatter<0, ...>) of
the same rank-one update, i.e. on the second stream while the next panel group was being streamed.
usage: overlap_from_trace.py <dir-or-kernel_trace.csv>"""
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
passes = [(a, b) for a, b, n, _ in rows if "k_scatter<0" in n or "k_scatter<(mfx::ScatterMode)0" in n]
def under(a, b):
    tot = 0
    for pa, pb in passes:
        lo, hi = max(a, pa), min(b, pb)
        if hi > lo:
            tot += hi - lo
    return tot
groups = {}
for a, b, n, q in rows:
    key = ("k_scatter_combine" if "k_scatter_combine" in n else "k_finalize" if "k_finalize" in n else
           "rccl" if ("rccl" in n.lower() or "nccl" in n.lower() or "oneRankReduce" in n) else None)
    if not key:
        continue
    g = groups.setdefault(key, [0, 0, 0, set()])
    g[0] += 1; g[1] += b - a; g[2] += under(a, b); g[3].add(q)
streams = sorted({q for _, _, n, q in rows if "k_scatter<" in n})
print(f"column passes (k_scatter<0>): {len(passes)} launches, mean {sum(b - a for a, b in passes) / max(1, len(passes)) / 1e3:.1f} us, streams/queues {streams}")
for key, (n, dur, ov, qs) in sorted(groups.items()):
    print(f"{key:20s} launches {n:6d}  total {dur / 1e3:10.1f} us  under a column pass {ov / 1e3:10.1f} us = {100.0 * ov / max(1, dur):5.1f} %  streams/queues {sorted(qs)}")

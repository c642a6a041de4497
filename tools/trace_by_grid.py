#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, template arguments, grid): count, average and total duration.
usage: trace_by_grid.py trace.csv [min_total_us] [name filter]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if flt and flt not in n:
        continue
    m = re.search(r"(k_[a-z0-9_]+)(I[A-Za-z0-9_]*E)?", n)
    short = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    wg = int(r["Workgroup_Size_X"])
    key = (short[:70], int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print(f"{len(rows)} dispatches, {tot:.1f} us in the selection")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] >= lim:
        print(f"{k[0]:70s} grid=({k[1]},{k[2]},{k[3]}) n={v[0]:5d} avg {v[1]/v[0]:7.2f} us total {v[1]:9.1f} us {100*v[1]/tot:5.1f}%")

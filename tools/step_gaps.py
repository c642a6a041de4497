#!/usr/bin/env python3
"""Idle gaps between consecutive kernels inside the timed guided steps of a rocprofv3 kernel-trace CSV of bench.py:
total idle per step, and the largest gaps with the kernels on either side (where the host is the bottleneck)."""
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_ddim_cfg" in r[2]]
segs = [rows[a + 1: b + 1] for a, b in zip(marks[:-1], marks[1:])]
first_spin = next((i for i, x in enumerate(segs) if any("spin_kernel" in r[2] for r in x)), len(segs))
segs = [x for x in segs[:first_spin] if len(x) > 1800][-6:]
def nm(n):
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:36]
for seg in segs[-3:]:
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i + 1][0] - seg[i][1], nm(seg[i][2]), nm(seg[i + 1][2])) for i in range(len(seg) - 1)]
    big = sorted(gaps, reverse=True)[:12]
    small = sum(g for g, _, _ in gaps if g <= 3000)
    print(f"step: {len(seg)} kernels span {span/1e3:.1f} us busy {busy/1e3:.1f} us idle {(span-busy)/1e3:.1f} us; gaps > 3 us: {sum(1 for g,_,_ in gaps if g > 3000)} totalling {sum(g for g,_,_ in gaps if g > 3000)/1e3:.1f} us; gaps <= 3 us total {small/1e3:.1f} us")
    for g, a, b in big:
        print(f"    {g/1e3:8.1f} us  after {a:28s} before {b}")

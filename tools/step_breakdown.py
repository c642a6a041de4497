#!/usr/bin/env python3
"""Per-(kernel, grid) breakdown of the LAST guided step in a rocprofv3 kernel-trace CSV of bench.py.
A guided step ends with k_ddim_cfg_step; the segment between the last two of them is one full step."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_ddim_cfg" in r[2]]
# one guided step of the TIMED region (full launch count, no device-side spin of the event-bracket pass)
segs = [rows[a + 1: b + 1] for a, b in zip(marks[:-1], marks[1:])]
first_spin = next((i for i, x in enumerate(segs) if any("spin_kernel" in r[2] for r in x)), len(segs))
segs = [x for x in segs[:first_spin] if len(x) > 1800]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = segs[-k]
span = (seg[-1][1] - seg[0][0]) / 1e3
busy = sum(e - s for s, e, *_ in seg) / 1e3
print(f"kernels {len(seg)} span {span:.1f} us busy {busy:.1f} us")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, gx, gy, gz in seg:
    nm = n.replace("_ZN2dh", "").replace("dh::", "").split("EEv")[0].split("(")[0][:44]
    agg[(nm, gx, gy, gz)][0] += 1
    agg[(nm, gx, gy, gz)][1] += (e - s) / 1e3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for kk, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{kk[0]:46s} grid=({kk[1]},{kk[2]},{kk[3]}) n={v[0]:4d} total {v[1]:8.1f} us avg {v[1]/v[0]:6.1f}  {v[1]/busy*100:4.1f}%")

#!/usr/bin/env python3
"""Print the kernel sequence of one engine pass from a rocprofv3 kernel-trace db (time_unet.py run)."""
import sqlite3, sys, collections
db = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
# a forward pass starts with k_f32_to_t (text) ; find the last complete B=1 forward: between two consecutive k_f32_to_t
idx = [i for i, r in enumerate(rows) if 'k_f32_to_t' in r[0]]
a, b = idx[-12], idx[-11]
seg = rows[a:b]
t0 = seg[0][1]
agg = collections.defaultdict(lambda: [0, 0.0])
prev_end = seg[0][1]
tot_gap = 0
for n, s, e, gx, gy, gz, wx in seg:
    k = n.replace('_ZN2dh', '').split('EEv')[0].split('(')[0][:40]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    tot_gap += max(0, s - prev_end); prev_end = e
print(f"segment kernels {len(seg)} span {(seg[-1][2]-t0)/1e3:.1f} us busy {sum(v[1] for v in agg.values()):.1f} us gaps {tot_gap/1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} n={v[0]:4d} total {v[1]:8.1f} us avg {v[1]/v[0]:6.1f}")
print("--- 25 longest kernels")
for n, s, e, gx, gy, gz, wx in sorted(seg, key=lambda r: -(r[2]-r[1]))[:25]:
    print(f"{(e-s)/1e3:7.1f} us grid=({gx//max(wx,1)},{gy},{gz}) {n.replace('_ZN2dh','').split('EEv')[0][:60]}")

#!/usr/bin/env python3
"""Per-kernel-type totals of the LAST guided step(s) in a rocprofv3 kernel-trace CSV of bench.py."""
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_ddim_cfg" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# guided steps of the TIMED region: segments between two DDIM steps that hold the full 7-pass launch count and no
# device-side spin (the spin marks the eager event-bracket pass); the last k of them
segs = [rows[a + 1: b + 1] for a, b in zip(marks[:-1], marks[1:])]
# guided steps of the TIMED region: full 7-pass segments (> 1800 launches; an unguided step has ~600) in front of the first device-side spin
# (the spin marks the eager event-bracket pass that follows the timed region); the schedule has a period of three
# steps with different launch counts, so whole periods are averaged
first_spin = next((i for i, x in enumerate(segs) if any("spin_kernel" in r[2] for r in x)), len(segs))
segs = [x for x in segs[:first_spin] if len(x) > 1800]
k = 3 * max(1, min(k, len(segs) // 3))
seg = [r for x in segs[-k:] for r in x]
busy = sum(e - s for s, e, _ in seg) / 1e3 / k
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    m = re.search(r"k_[a-z0-9_]+", n)
    nm = m.group(0) if m else n[:40]
    if nm.startswith("k_gemm_dma"):
        nm = "k_gemm_dma"
    agg[nm][0] += 1
    agg[nm][1] += (e - s) / 1e3
print(f"per step: kernels {len(seg)//k} busy {busy:.1f} us")
for kk, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{kk:28s} n/step={v[0]/k:7.1f} us/step {v[1]/k:8.1f} avg {v[1]/v[0]:6.1f}  {v[1]/k/busy*100:4.1f}%")

#!/usr/bin/env python3
"""Per-kernel-type totals of the LAST guided step(s) in a rocprofv3 kernel-trace CSV of bench.py."""
import csv, sys, collections, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_ddim_cfg" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # the last steps are the eager event-bracket pass
seg = rows[marks[-k - 1 - skip] + 1: marks[-1 - skip] + 1]
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

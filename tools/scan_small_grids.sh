#!/bin/bash
# every launch shape of the guided steps with its workgroup count (rocprofv3 kernel trace of bench.py's timed region): the ones that
# leave most of the 256 CUs idle and still take real time are the candidates for a different block shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/scan; rocprofv3 --kernel-trace --output-format csv -d /tmp/scan -- python3 bench.py --no-phases --no-res768 --no-cpu-baseline --batch-edits 0 > /dev/null 2>&1
f=$(ls /tmp/scan/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY' > gpurun_out/scan_small_grids.txt
import csv, sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"(k_[a-z0-9_]+)(I[A-Za-z0-9_]*E)?", n)
    short = (m.group(1) + (m.group(2) or "")) if m else n[:50]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    k = (short[:64], g, wg)
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot/1e3:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    waves = k[1] * (k[2] // 64)
    if waves < 1024 and v[1] / tot > 0.001:
        print(f"{100*v[1]/tot:5.2f}%  n={v[0]:6d} avg {v[1]/v[0]:7.2f} us  workgroups {k[1]:5d} x {k[2]:4d} threads = {waves:5d} waves  {k[0]}")
PY
cat gpurun_out/scan_small_grids.txt | head -50

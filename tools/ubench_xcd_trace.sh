#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of the reader kernel of tools/ubench_xcd.hip by chunk size and reader shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/xcdtr; rocprofv3 --kernel-trace --output-format csv -d /tmp/xcdtr -- tools/bin/ubench_xcd > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/xcdtr/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rd = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'k_r' in r['Kernel_Name']]
i = 0
for kb in (4, 16, 64, 128):
    for sr in (0, 8, 1, 3, 128):
        seg = sorted(rd[i + 5:i + 55]); i += 55
        print(f"chunk {kb:4d} KB ({kb*256/1024:5.1f} MB total) shift {sr:3d} ({'same' if sr % 8 == 0 else 'other'} XCD): reader kernel median {seg[len(seg)//2]:6.2f} us  min {seg[0]:6.2f}")
PY

#!/bin/bash
# kernel accounting of the per-image phase (null-text inversion + initial inference): rocprofv3 kernel trace of tools/time_invert.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/invert
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/invert -- python3 tools/time_invert.py > gpurun_out/invert/log.txt 2>&1
f=$(ls gpurun_out/invert/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 20000 > gpurun_out/invert/by_grid.txt
python3 - "$f" > gpurun_out/invert/by_type.txt <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"k_[a-z0-9_]+",r["Kernel_Name"]); nm=m.group(0) if m else r["Kernel_Name"][:40]
    agg[nm][0]+=1; agg[nm][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
print(f"total busy {tot/1e3:.1f} ms")
for k,v in sorted(agg.items(),key=lambda kv:-kv[1][1])[:30]: print(f"{k:28s} n={v[0]:7d} total {v[1]/1e3:9.1f} ms avg {v[1]/v[0]:7.1f} us {100*v[1]/tot:5.1f}%")
PY
rm -f $f
grep "^rep" gpurun_out/invert/log.txt

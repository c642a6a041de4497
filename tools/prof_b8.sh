#!/bin/bash
set -eu
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
mkdir -p gpurun_out/b8 gpurun_out/l96
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b8 -- python3 tools/time_unet.py 8 > gpurun_out/b8/log.txt 2>&1
f=$(ls gpurun_out/b8/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 500 > gpurun_out/b8/by_grid.txt
python3 - "$f" > gpurun_out/b8/by_type.txt <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m=re.search(r"k_[a-z0-9_]+",r["Kernel_Name"]); nm=m.group(0) if m else r["Kernel_Name"][:40]
    agg[nm][0]+=1; agg[nm][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
for k,v in sorted(agg.items(),key=lambda kv:-kv[1][1])[:30]: print(f"{k:28s} n={v[0]:6d} total {v[1]:10.1f} avg {v[1]/v[0]:7.1f} {100*v[1]/tot:5.1f}%")
PY
rm -f $f
DH_LATENT=96 DH_DTYPE=bf16 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/l96 -- python3 tools/time_unet.py 1 > gpurun_out/l96/log.txt 2>&1
f=$(ls gpurun_out/l96/*/*kernel_trace.csv | head -1)
python3 tools/trace_by_grid.py $f 300 > gpurun_out/l96/by_grid.txt
rm -f $f
tail -3 gpurun_out/b8/log.txt gpurun_out/l96/log.txt

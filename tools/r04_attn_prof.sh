#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp
DH_TIMELINE_KERNEL=dq DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdh_stamp_dq.so python3 tools/attn_timeline_dkv.py 4096 5 1 2>&1 | grep "^dq" | tail -2
python3 -m pytest tests/test_unet_kernels_gpu.py -x -q -m gpu -k attention 2>&1 | tail -2
for lib in tools/bin/libdh_attn_d.so diffusionhandles_amd/libdiffhandles_hip.so tools/bin/libdh_attn_d.so diffusionhandles_amd/libdiffhandles_hip.so; do
  rm -rf /tmp/pa_$$;
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/pa_$$ -- python3 tools/bench_attn.py > /dev/null 2>&1
  echo "== $lib"
  python3 - /tmp/pa_$$ <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_attn[a-z_]+", r["Kernel_Name"])
        if not m: continue
        g = (m.group(0), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))
        agg[g][0] += 1; agg[g][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for g, v in sorted(agg.items()):
    if g[0] in ("k_attn_bwd_dq", "k_attn_fwd") and g[4] >= 512: print(f"{g[0]:18s} grid=({g[1]},{g[2]},{g[3]}) x {g[4]:4d} thr  n={v[0]:4d} avg {v[1]/v[0]:8.2f} us")
PY
done

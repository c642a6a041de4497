#!/bin/bash
# rocprofv3 kernel traces of the HBM-bound pieces (guidance energy, batched K=8 re-projection) -> achieved GB/s per kernel.
# Run through gpurun from the repo root; CSVs land in gpurun_out/hbm/ and are copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/hbm
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for res in 512 768; do
  rm -rf /tmp/prof_e$res /tmp/prof_r$res
  DH_RES=$res rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e$res -- python3 $R/tools/bench_energy.py 2>/dev/null | tail -1 > $O/energy_$res.json
  python3 $R/tools/hbm_report.py energy $(ls /tmp/prof_e$res/*/*kernel_trace.csv | head -1) $O/energy_$res.json > $O/energy_${res}_kernel_gbps.csv
  DH_RES=$res rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r$res -- python3 $R/tools/bench_reproject.py 2>/dev/null | tail -1 > $O/reproject_$res.txt
  cp $(ls /tmp/prof_r$res/*/*kernel_stats.csv | head -1) $O/reproject_k8_${res}_kernel_stats.csv
  NFG=$(python3 -c "import sys; sys.path.insert(0,'$R'); from diffusionhandles_amd.synthetic import make_scene; print(int(make_scene($res)[2].sum()))")
  python3 $R/tools/hbm_report.py reproject $(ls /tmp/prof_r$res/*/*kernel_trace.csv | head -1) $res $NFG 8 > $O/reproject_k8_${res}_kernel_gbps.csv
done
cat $O/energy_512_kernel_gbps.csv $O/reproject_k8_512_kernel_gbps.csv $O/reproject_512.txt

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --batch-edits 0 --profile-steps 1"
bash tools/lab.sh ab "$BENCH" diffusionhandles_amd/libdiffhandles_hip.so tools/bin/libdh_two.so > gpurun_out/r04_ab_two_per_cu.txt 2>&1
for lib in diffusionhandles_amd/libdiffhandles_hip.so tools/bin/libdh_two.so diffusionhandles_amd/libdiffhandles_hip.so tools/bin/libdh_two.so; do
  echo "== $lib" >> gpurun_out/r04_ab_two_per_cu.txt
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/time_unet.py 2 2>&1 | grep "^B=" >> gpurun_out/r04_ab_two_per_cu.txt
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_LATENT=96 DH_DTYPE=bf16 python3 tools/time_unet.py 1,2 2>&1 | grep "^B=" | sed 's/^/L96 /' >> gpurun_out/r04_ab_two_per_cu.txt
done
bash tools/lab.sh refresh > gpurun_out/r04_refresh.log 2>&1
python3 tools/step_by_level.py gpurun_out/final/step_breakdown_by_grid.txt > gpurun_out/final/step_by_level.txt

#!/bin/bash
# round-4 GPU run 1: GEGLU-epilogue parity + same-box A/B against the round-3 library
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
python3 -m pytest tests/test_unet_kernels_gpu.py -x -q -m gpu -k "geglu or gemm_dense" 2>&1 | tail -15 > gpurun_out/r04_run1_kernels.txt
python3 -m pytest tests/test_unet_engine_gpu.py -x -q -m gpu -s 2>&1 | grep -E "rel err|d_text|d_sample|passed|failed|Error|assert" | tail -60 > gpurun_out/r04_run1_engine.txt
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-res768 --batch-edits 0 --profile-steps 1"
bash tools/ab.sh "$BENCH" tools/bin/libdh_r03.so diffusionhandles_amd/libdiffhandles_hip.so > gpurun_out/r04_run1_ab.txt 2>&1
bash tools/ab_libs.sh tools/bin/libdh_r03.so diffusionhandles_amd/libdiffhandles_hip.so > gpurun_out/r04_run1_ab_unet.txt 2>&1

#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
python3 -m pytest tests/test_unet_kernels_gpu.py tests/test_unet_engine_gpu.py -x -q -m gpu -k "gemm_dense or batch8" 2>&1 | tail -5 > gpurun_out/r04_run5_tests.txt
bash tools/lab.sh evidence > gpurun_out/r04_evidence.log 2>&1

#!/bin/bash
# k_gemm_dma ablations (0 = full, 1 = staging only: no LDS reads / MFMA, 2 = compute only: no DMA in the loop) on the batch-8 shapes
cd $GRAFT_REPO_ROOT
for abl in 0 1 2; do
  echo "ABLATE $abl"
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdiffhandles_hip_tuning.so DH_DBG_PRETILED=1 DH_SHAPES=${1:-b8} DH_GEMM_ABLATE=$abl python3 tools/bench_gemm.py 2>&1 | grep "^M="
done

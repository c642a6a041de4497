#!/bin/bash
# same-box A/B of library builds: tools/ab_libs.sh lib1 lib2 ... ; U-Net passes at B = 1, 2, 8, the 96x96 latent (bf16), the batch-8 GEMM shapes
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/time_unet.py 1,2,8 2>&1 | grep "^B="
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_LATENT=96 DH_DTYPE=bf16 python3 tools/time_unet.py 1 2>&1 | grep "^B=" | sed 's/^/L96 /'
  done
done
for lib in "$@"; do
  echo "== $lib"
  DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_DBG_PRETILED=1 DH_SHAPES=b8 python3 tools/bench_gemm.py 2>&1 | grep "^M="
done

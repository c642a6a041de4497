#!/bin/bash
# A/B of the GEMM dispatch knobs on one box (run through gpurun): per-pass U-Net time at batch 1 for each setting
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "== $*"; env "$@" timeout 120 python tools/time_unet.py ${BATCHES:-1} 2>&1 | grep "B="; }
run X=0
run DH_SPLITK_TARGET=128
run DH_SPLITK_TARGET=160
run DH_SPLITK_TARGET=192
run DH_SPLITK_TARGET=224
run DH_SPLITK_TARGET=192 DH_BIG_TILES=96
run DH_SPLITK_TARGET=192 DH_SPLITK_MINKT=32
run DH_SPLITK_TARGET=192 DH_BIG_TILES=96 DH_SPLITK_MINKT=32
run DH_SPLITK_TARGET=160 DH_BIG_TILES=96
run X=0
run DH_SPLITK_TARGET=192

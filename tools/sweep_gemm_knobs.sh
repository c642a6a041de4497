#!/bin/bash
# A/B of dispatch knobs on one box (run through gpurun), judged on the guided step of bench.py (policy changes that win
# on the repeated full pass of tools/time_unet.py have lost here)
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "== $*: $(env "$@" timeout 200 python bench.py --no-time-edit --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run X=0
run DH_SPLITK_TARGET=288
run DH_SPLITK_TILES=160
run DH_SPLITK_TILES=256
run DH_GEMM_MW=48
run DH_GEMM_MW=96
run DH_GEMM_MANY=256
run DH_GEMM_MANY=1024
run DH_KG2_MINKT=24
run DH_GEMM_MW128=0
run X=0

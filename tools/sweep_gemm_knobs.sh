#!/bin/bash
# A/B of dispatch knobs on one box (run through gpurun), judged on the guided step of bench.py (policy changes that win
# on the repeated full pass of tools/time_unet.py have lost here)
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "== $*: $(env "$@" timeout 200 python bench.py --no-time-edit --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run X=0
run DH_BIG_TILES=96
run DH_BIG_TILES=200
run DH_SPLITK_MINKT=28
run DH_SPLITK_MINKT=40
run DH_SPLITK_TARGET=240
run DH_SPLITK_TILES=128
run DH_GN_SLICES=24
run DH_GN_SLICES=48
run DH_ATTN_KS=2
run DH_ATTN_KS=4
run DH_ATTN_QW=2
run DH_ATTN_QW=4
run X=0

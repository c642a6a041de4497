#!/bin/bash
# A/B of dispatch knobs on one box (run through gpurun), judged on the guided step of bench.py (policy changes that win
# on the repeated full pass of tools/time_unet.py have lost here)
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "== $*: $(env "$@" timeout 200 python bench.py --no-time-edit --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
run X=0
run DH_NARROW_TILES=40 DH_KG2_MINKT=8
run DH_NARROW_TILES=40 DH_KG2_MINKT=4
run DH_NARROW_TILES=64 DH_KG2_MINKT=8
run DH_NARROW_TILES=48 DH_KG2_MINKT=6
run DH_NARROW_TILES=0 DH_KG2_MINKT=8
run DH_NARROW_TILES=40 DH_KG2_MINKT=12
run X=0

#!/bin/bash
# split-K target / two-per-CU knobs of the tuning build on the guided step (bench.py) and on the B = 1 / B = 2 / 96x96 passes
set -u
cd "${GRAFT_REPO_ROOT:?}"
export DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdiffhandles_hip_tuning.so
run() { echo "== $*: $(env "$@" timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-phases --no-res768 --batch-edits 0 --profile-steps 1 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; }
for rep in 1 2; do
run X=0
run DH_SPLITK_TARGET=384
run DH_SPLITK_TARGET=512
run DH_SPLITK_TARGET=512 DH_SPLITK_TILES=256
run DH_GEMM_TWO_PER_CU=0
done

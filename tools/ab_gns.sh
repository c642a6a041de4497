#!/bin/bash
# same-box A/B of two library builds on the guided step (tools/ab_inplace.py-style timing, in-place I/O on): interleaved repeats
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x"]
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "ab_inplace.py")).read().split("finals = {}")[0])
gd._inplace_io = True
run(6)
a, _ = run(76)
print(f"guided steps/s {a:.2f}")
PY
  done
done

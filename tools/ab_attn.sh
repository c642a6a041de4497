#!/bin/bash
# same-box A/B of attention builds: tools/ab_attn.sh lib1 lib2 ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/bench_attn.py 2>&1 | grep "^B="
  done
done

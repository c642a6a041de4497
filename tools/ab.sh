#!/bin/bash
# same-box A/B of library builds: tools/ab.sh "<command>" libA.so libB.so ...   (each run twice, interleaved)
CMD=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib bash -c "$CMD" 2>&1 | grep -E "B=|steps/s|value" | cut -c1-160
  done
done

#!/bin/bash
# round-4 GPU run 3: the whole GPU suite on the current build + wider lane counts
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_run3_tests.txt
DH_LANES_CASES=wide timeout 900 python3 tools/bench_lanes.py > gpurun_out/r04_lanes_wide.txt 2>&1

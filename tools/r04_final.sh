#!/bin/bash
# the round's closing run: whole GPU suite, smoke, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/final
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/final/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
timeout 900 python3 bench.py 2> gpurun_out/final/bench_n1.err | tail -1 > gpurun_out/final/bench_n1.json

#!/bin/bash
# round-4 GPU run 2: lanes parity + throughput
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out
python3 -m pytest tests/test_loops_gpu.py -x -q -m gpu -k "lanes or batched_edits or teacher_forced or guided_inference_matches or initial_inference" 2>&1 | tail -15 > gpurun_out/r04_run2_tests.txt
timeout 900 python3 tools/bench_lanes.py > gpurun_out/r04_lanes.txt 2>&1

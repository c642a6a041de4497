#!/bin/bash
# the round's closing run: whole GPU suite, smoke, the default bench line, then the evidence the profiles/ files come from
set -u
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/final
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/final/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
bash tools/lab.sh evidence > gpurun_out/final/evidence.log 2>&1
bash tools/lab.sh prof-b8 > gpurun_out/final/prof_b8.log 2>&1
DH_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --no-cpu-baseline 2> gpurun_out/final/bench_gloo2_one_gpu.err | tail -1 > gpurun_out/final/bench_gloo2_one_gpu.json

#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DH_DBG_PRETILED=1
rm -rf /tmp/warmth; rocprofv3 --kernel-trace --output-format csv -d /tmp/warmth -- python3 tools/bench_gemm_warmth.py run > /tmp/warmth.log 2>&1
tail -3 /tmp/warmth.log | cut -c1-200
python3 tools/bench_gemm_warmth.py parse /tmp/warmth | tee gpurun_out/gemm_warmth.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdh_stamp.so python3 tools/attn_timeline_dkv.py 4096 5 1 2>&1 | grep "^N="
python3 -m pytest tests/test_unet_kernels_gpu.py -x -q -m gpu -k attention 2>&1 | tail -2
bash tools/r04_attn_prof.sh

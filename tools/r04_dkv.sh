#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdh_stamp.so python3 tools/attn_timeline_dkv.py 4096 5 1 2>&1 | grep "^N="
python3 -m pytest tests/test_unet_kernels_gpu.py -x -q -m gpu -k attention 2>&1 | tail -2
for rep in 1 2; do for lib in tools/bin/libdh_before_dkv.so diffusionhandles_amd/libdiffhandles_hip.so; do
  echo "== $lib"; DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/bench_attn.py 2>&1 | grep "^B="
done; done

#!/bin/bash
# same-box A/B: conv K order (tools/bin/libdh_tapmajor.so = -DDH_CONV_TAP_MAJOR) on U-Net passes at B = 1, 8 and the 96x96 latent, then the guided step
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in tools/bin/libdh_tapmajor.so diffusionhandles_amd/libdiffhandles_hip.so; do
    echo "== $lib"
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib python3 tools/time_unet.py 1,2,8 2>&1 | grep "^B="
    DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/$lib DH_LATENT=96 DH_DTYPE=bf16 python3 tools/time_unet.py 1 2>&1 | grep "^B=" | sed 's/^/L96 /'
  done
done
bash tools/ab_gns.sh tools/bin/libdh_tapmajor.so diffusionhandles_amd/libdiffhandles_hip.so

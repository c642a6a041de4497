#!/bin/bash
# Tuning build of the library (-DDH_TUNING: timing-only ablation knobs compiled in) into tools/bin/ (git-ignored, travels
# with gpurun).  Select it with DIFFHANDLES_LIB=tools/bin/libdiffhandles_hip_tuning.so; the product library never has the knobs.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${DH_SRC:-$ROOT/diffusionhandles_amd/csrc}          # DH_SRC: another checkout's csrc (A/B of two source states on one box)
NAME=${DH_NAME:-libdiffhandles_hip_tuning.so}
OBJ=${TMPDIR:-/tmp}/dh_obj_$NAME
mkdir -p "$OBJ" "$ROOT/tools/bin"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 ${DH_DEFS--DDH_TUNING} -Wno-unused-function -Wno-unused-result"
pids=()
for f in api.cpp geometry.hip mesh.hip cells.hip energy.hip loop_ops.hip gemm.hip attention.hip unet_kernels.hip unet_engine.cpp vae_engine.cpp text_engine.cpp debug_api.cpp; do
  extra=""; case $f in geometry.hip|mesh.hip) extra="-ffp-contract=off";; esac
  ( cd "$SRC" && $HIPCC $FLAGS $extra -x hip -c $f -o "$OBJ/${f%.*}.o" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/$NAME" "$OBJ"/*.o
echo built "$ROOT/tools/bin/$NAME"

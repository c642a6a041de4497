#!/bin/bash
# Timing-only ablations of the attention forward loop: builds tools/bin/libdh_attn_<n>.so = the product objects with
# attention.hip recompiled under -DDH_ATTN_ABL=n (1 no exp2, 4 no PV MFMAs / transposed reads, 5 no QK MFMAs / K reads; the
# variants 2 no K/V refetch, 3 no barriers, 6 no refetch and no commit were measured on the single-buffered loop of round 2,
# profiles/r02_attention_fwd_ablation.txt, and went away with it).  Run tools/bench_attn.py with DIFFHANDLES_LIB pointing at each.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/diffusionhandles_amd/csrc
make -C "$SRC" -j8 >/dev/null
mkdir -p "$ROOT/tools/bin"
OBJS=$(ls "$SRC"/*.o | grep -v attention.o)
for n in ${@:-0 1 4 5}; do
  ( cd "$SRC" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDH_ATTN_ABL=$n -c attention.hip -o /tmp/dh_attn_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/libdh_attn_$n.so" $OBJS /tmp/dh_attn_$n.o ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la "$ROOT"/tools/bin/libdh_attn_*.so

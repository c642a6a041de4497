#!/bin/bash
# per-shape tile / split-K search on the GEMM shapes of the batch-8 pass and of the 768x768 (96x96 latent) pass
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export DIFFHANDLES_LIB=$GRAFT_REPO_ROOT/tools/bin/libdiffhandles_hip_tuning.so
mkdir -p gpurun_out/sweep
DH_GEMM_LOG=1 python3 tools/time_unet.py 8 2> gpurun_out/sweep/gemmlog_b8.txt | grep "^B="
DH_GEMM_LOG=1 DH_LATENT=96 python3 tools/time_unet.py 1 2> gpurun_out/sweep/gemmlog_l96.txt | grep "^B="
for mode in b8 l96; do
  if [ $mode = b8 ]; then export DH_SWEEP_BATCH=8; else export DH_SWEEP_BATCH=1; fi
  rm -rf gpurun_out/sweep/trace_$mode
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sweep/trace_$mode -- python3 tools/sweep_gemm_shapes.py run gpurun_out/sweep/gemmlog_$mode.txt gpurun_out/sweep/manifest_$mode.json > gpurun_out/sweep/run_$mode.log 2>&1
  f=$(ls gpurun_out/sweep/trace_$mode/*/*kernel_trace.csv | head -1)
  python3 tools/sweep_gemm_shapes.py parse gpurun_out/sweep/manifest_$mode.json $f > gpurun_out/sweep/result_$mode.txt 2>&1
  rm -rf gpurun_out/sweep/trace_$mode
  tail -4 gpurun_out/sweep/result_$mode.txt
done

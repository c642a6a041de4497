#!/bin/bash
# Re-collect the judged measurements on the GPU box (run through gpurun from the repo root); summaries land in
# gpurun_out/final/ and are copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --no-phases --no-res768 --no-cpu-baseline --batch-edits 0 2>/dev/null | tail -1 > $O/bench_under_rocprof.json
cd $R
T=$(ls /tmp/prof_bench/*/*kernel_trace.csv | head -1)
S=$(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1)
cp $S $O/rocprofv3_kernel_stats.csv
python tools/step_types.py $T 3 > $O/step_kernel_types.txt
python tools/step_breakdown.py $T 3 > $O/step_breakdown_by_grid.txt
python tools/step_gaps.py $T > $O/step_gaps.txt
head -3 $O/bench_n1.json | cut -c1-400
head -12 $O/step_kernel_types.txt

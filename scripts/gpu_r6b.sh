#!/bin/bash
# round 6: kernel timeline of the sort family
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6b/prof -o sort -- python scripts/prof_sort.py sort sort_values > gpurun_out/r6b/prof.log 2>&1
tail -3 gpurun_out/r6b/prof.log
find gpurun_out/r6b/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r6b/kernel_stats.csv
python scripts/kstats.py gpurun_out/r6b/kernel_stats.csv 2>/dev/null | head -30 || head -30 gpurun_out/r6b/kernel_stats.csv
find gpurun_out/r6b/prof -name '*kernel_trace.csv' | head -1 | xargs -I{} cp {} gpurun_out/r6b/kernel_trace.csv
rm -rf gpurun_out/r6b/prof

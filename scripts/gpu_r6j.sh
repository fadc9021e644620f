#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6j
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6j/prof -o t -- python scripts/prof_sort_rmat.py > gpurun_out/r6j/prof.log 2>&1
f=$(find gpurun_out/r6j/prof -name '*kernel_stats.csv' | head -1)
python scripts/kstats.py $f | grep "tsamd\|kernel " | head
f=$(find gpurun_out/r6j/prof -name '*kernel_trace.csv' | head -1)
python scripts/trace_timeline.py $f 8
rm -rf gpurun_out/r6j/prof

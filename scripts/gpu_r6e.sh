#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
for what in construct coalesce; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6e/prof_$what -o t -- python scripts/prof_sort.py $what > gpurun_out/r6e/prof_$what.log 2>&1
f=$(find gpurun_out/r6e/prof_$what -name '*kernel_trace.csv' | head -1)
echo "== $what"; python scripts/trace_timeline.py $f ${1:-22}
rm -rf gpurun_out/r6e/prof_$what
done

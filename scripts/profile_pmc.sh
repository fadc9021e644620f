#!/bin/bash
# PMC evidence for every kernel family, on the GPU box through gpurun:
#   pass 1/2: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over
#             scripts/prof_kernels.py
#   pass 3:   rocprofv3 --kernel-trace --stats of the same script (durations of the same launches)
# Output: gpurun_out/$1/ ; condensed into profiles/ by scripts/summarize_pmc.py.
set -u
TAG=${1:-r03_pmc}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-include-regex "tsamd" --output-format csv -d $OUT/pmc_$C -o pmc -- python scripts/prof_kernels.py "$@" > $OUT/pmc_$C.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o k -- python scripts/prof_kernels.py "$@" > $OUT/trace.log 2>&1
find $OUT -name "*agent_info*" -delete
ls -R $OUT | head -40
tail -2 $OUT/trace.log

#!/bin/bash
# Round-4 evidence on the GPU box (through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --headline-only --no-pmc` (every launch of the dominant kernel
#      is a north-star launch: its average must agree with the HIP-event time of the bench line)
#   2. the plain default `bench.py` line of the same box (it measures its own fabric traffic in --pmc children)
# Output: gpurun_out/$1/ ; condensed into profiles/ by scripts/kstats.py.
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o bench -- python bench.py --headline-only --no-pmc > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
rm -f $OUT/trace_headline/*/*kernel_trace.csv $OUT/trace_headline/*kernel_trace.csv
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
find $OUT -name "*agent_info*" -delete
tail -3 $OUT/bench.time
ls $OUT $OUT/trace_headline | head -20

#!/bin/bash
# Round-3 evidence on the GPU box (through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --headline-only` (every launch of the dominant kernel is a
#      north-star launch: its average must agree with the HIP-event time of the bench line)
#   2. the plain default `bench.py` line of the same box
#   3. the PMC passes over every kernel family (scripts/profile_pmc.sh)
# Output: gpurun_out/$1/ ; condensed by scripts/summarize_round3.py and scripts/summarize_pmc.py.
set -u
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o bench -- python bench.py --headline-only > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
rm -f $OUT/trace_headline/*/*kernel_trace.csv $OUT/trace_headline/*kernel_trace.csv
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
bash scripts/profile_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
find $OUT -name "*agent_info*" -delete
tail -3 $OUT/bench.time
ls $OUT $OUT/trace_headline | head -20

#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (headline + control + C2/C3/C4)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a few north-star SpMM launches, drop-in op and
#      relabelled layout
#   3. the plain bench.py line of the same box
# Output lands in gpurun_out/$1/ ; summaries are condensed into profiles/ by scripts/summarize_profile.py.
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py > $OUT/bench_under_rocprof.log 2>&1
# the same command restricted to the headline workload: every launch of the dominant kernel is a north-star launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o bench -- python bench.py --headline-only > $OUT/bench_headline_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "spmm_" --output-format csv -d $OUT/pmc_$C -o pmc -- python scripts/prof_spmm.py 21 128 sum 3 > $OUT/pmc_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-include-regex "spmm_" --output-format csv -d $OUT/pmcrel_$C -o pmc -- python scripts/prof_spmm_relabelled.py > $OUT/pmcrel_$C.log 2>&1
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -f $OUT/trace*/*/*kernel_trace.csv $OUT/trace*/*kernel_trace.csv  # large; the stats are what is kept
ls -R $OUT | head -60

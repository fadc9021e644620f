#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a few SpMM launches
# Output lands in gpurun_out/$1/ ; summaries are condensed into profiles/ by scripts/summarize_profile.py.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 20 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "spmm_" --output-format csv -d $OUT/pmc_$C -o pmc -- python scripts/prof_spmm.py 21 128 sum 3 > $OUT/pmc_$C.log 2>&1
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload c2 --no-cpu-baseline > $OUT/bench_c2.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline --steps 20 > $OUT/bench_c5.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_extra -o extra -- python scripts/bench_extra.py c3 c4 coalesce vbw > $OUT/bench_extra_under_rocprof.log 2>&1
python scripts/bench_extra.py c3 c4 coalesce vbw narrow 2>/dev/null | grep "^{" > $OUT/bench_extra.jsonl
ls -R $OUT | head -40

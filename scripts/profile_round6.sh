#!/bin/bash
# Round-6 evidence on the GPU box (through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --headline-only --no-pmc` (every launch of the dominant kernel
#      is a north-star launch: its average must agree with the HIP-event time of the bench line)
#   2. the plain default `bench.py` run of the same box: stdout = the ONE compact line the driver parses,
#      profiles/bench_last_full.json = the full result
#   3. the SpSpMM stress row (behind --stress in bench.py) through the same row function
# Output: gpurun_out/$1/ ; condensed into profiles/ by scripts/kstats.py.
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o bench -- python bench.py --headline-only --no-pmc > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
rm -f $OUT/trace_headline/*/*kernel_trace.csv $OUT/trace_headline/*kernel_trace.csv
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
cp profiles/bench_last_full.json $OUT/bench_full.json
timeout 300 python - > $OUT/stress_row.json 2> $OUT/stress_row.err <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
import pytorch_sparse_amd
from tests import baseline_configs as bc
torch.set_num_threads(32)
print(json.dumps(bc.run_spspmm(torch.device('cuda:0'), 'stress', cpu=False, iters=5)))
PY
find $OUT -name "*agent_info*" -delete
tail -3 $OUT/bench.time
wc -c $OUT/bench.json
cat $OUT/stress_row.json | cut -c1-400

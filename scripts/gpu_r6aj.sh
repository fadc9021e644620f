#!/bin/bash
# round 6 (re-entry): look-back that widens after a first round without a prefix: tests, A/B, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6aj
mkdir -p $OUT
python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py -m gpu -x -q 2>&1 | tail -3
bash scripts/ab_sort_r6.sh r6aj/ab.jsonl --big
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o co -- python scripts/prof_sort.py coalesce > $OUT/prof.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python scripts/kstats.py $f | head -6 | tee $OUT/coalesce_kernel_stats.txt
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv

#!/bin/bash
# round 6 (re-entry): fused coalesce reduction + one fill: tests, same-box timing, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ad
mkdir -p $OUT
python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py tests/test_cabi.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  python scripts/bench_sort_quick.py fused >> $OUT/ab.jsonl 2>> $OUT/ab.err
  TSAMD_COALESCE_UNFUSED=1 python scripts/bench_sort_quick.py unfused >> $OUT/ab.jsonl 2>> $OUT/ab.err
done
cat $OUT/ab.jsonl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o co -- python scripts/prof_sort.py coalesce > $OUT/prof.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python scripts/kstats.py $f | head -16 | tee $OUT/coalesce_kernel_stats.txt
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv

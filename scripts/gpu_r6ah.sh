#!/bin/bash
# round 6 (re-entry): the contiguous result of a coalesce with duplicates placed on the device before the host sync
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ah
mkdir -p $OUT
python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py tests/test_random_cases_gpu.py tests/test_jit.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  python scripts/bench_sort_quick.py placed >> $OUT/ab.jsonl 2>> $OUT/ab.err
  TSAMD_COALESCE_UNFUSED=1 python scripts/bench_sort_quick.py unfused >> $OUT/ab.jsonl 2>> $OUT/ab.err
done
cat $OUT/ab.jsonl

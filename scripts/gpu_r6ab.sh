#!/bin/bash
# round 6 (re-entry): reduction of the duplicates' values fused into the compacting bucket sort -- tests, then same-box timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6ab
python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py -m gpu -x -q -k "coalesce or transpose or sort" 2>&1 | tail -8
for rep in 1 2 3; do
  python scripts/bench_sort_quick.py fused >> gpurun_out/r6ab/ab.jsonl 2>> gpurun_out/r6ab/ab.err
  TSAMD_COALESCE_UNFUSED=1 python scripts/bench_sort_quick.py unfused >> gpurun_out/r6ab/ab.jsonl 2>> gpurun_out/r6ab/ab.err
done
cat gpurun_out/r6ab/ab.jsonl
tail -3 gpurun_out/r6ab/ab.err

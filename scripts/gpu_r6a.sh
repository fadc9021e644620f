#!/bin/bash
# round 6, first contact of the bucket sort: sort tests, then the sort bench (7.5 M and 75 M entries)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sort_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r6a_pytest_sort.log
cat gpurun_out/r6a_pytest_sort.log
timeout 600 python scripts/bench_sort.py --big > gpurun_out/r6a_bench_sort.jsonl 2> gpurun_out/r6a_bench_sort.err
cat gpurun_out/r6a_bench_sort.jsonl; tail -3 gpurun_out/r6a_bench_sort.err

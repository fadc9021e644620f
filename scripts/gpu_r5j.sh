#!/bin/bash
mkdir -p gpurun_out/r5j
echo "== 7.5M"; timeout 120 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5j/sort_group_7m5.jsonl
echo "== 75M"; E=75000000 MN=4194304 timeout 200 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5j/sort_group_75m.jsonl
( time timeout 300 python -m pytest tests/test_sort_gpu.py tests/test_api_gpu.py -m gpu -x -q -k "sort or coalesce or transpose or construct" ) 2>&1 | tail -5

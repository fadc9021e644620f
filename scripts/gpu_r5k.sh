#!/bin/bash
mkdir -p gpurun_out/r5k
echo "== 7.5M"; timeout 120 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5k/sort_sleep_7m5.jsonl
echo "== 75M"; E=75000000 MN=4194304 timeout 200 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5k/sort_sleep_75m.jsonl

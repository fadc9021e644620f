#!/bin/bash
mkdir -p gpurun_out/r5i
echo "== 7.5M"; timeout 120 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5i/sort_look_7m5.jsonl
echo "== 75M"; E=75000000 MN=4194304 timeout 200 python scripts/exp_sort_variants.py 2>/dev/null | tee gpurun_out/r5i/sort_look_75m.jsonl

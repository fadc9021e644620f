#!/bin/bash
mkdir -p gpurun_out/r5al
R=$PWD
python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py tests/test_random_cases_gpu.py -q -m gpu -k "spspmm or random" 2>&1 | tail -2
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py c4 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5al/ab.jsonl; }
for rep in 1 2 3 4; do
  run shipped_hoist
  run old LD_PRELOAD=$R/build/ab/oldc4.so
done
cat gpurun_out/r5al/ab.jsonl

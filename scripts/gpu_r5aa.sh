#!/bin/bash
# SpSpMM register sort: lane ^ 8 (and ^ 4) exchanges on DPP instead of ds_swizzle -- same-box A/B (C4 + stress)
mkdir -p gpurun_out/r5aa
R=$PWD
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py c4 stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5aa/ab.jsonl; }
for rep in 1 2 3; do
  run shipped_xor8
  run noxor8 LD_PRELOAD=$R/build/ab/noxor8.so
  run xor8_xor4 LD_PRELOAD=$R/build/ab/xor4.so
done
cat gpurun_out/r5aa/ab.jsonl
python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py -q -m gpu -k "spspmm" 2>&1 | tail -2
LD_PRELOAD=$R/build/ab/xor4.so python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py -q -m gpu -k "spspmm" 2>&1 | tail -2

#!/bin/bash
# SpSpMM stress: 8 (shipped) vs 4 persistent workgroups per CU in the big-bin count kernel; 2^12-column ranges
mkdir -p gpurun_out/r5ae
R=$PWD
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ae/ab.jsonl; }
for rep in 1 2; do
  run shipped_count8
  run count4 LD_PRELOAD=$R/build/ab/count4.so
  run range12 LD_PRELOAD=$R/build/ab/range12.so
done
CHECK=1 VARIANT=shipped_check python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ae/ab.jsonl
cat gpurun_out/r5ae/ab.jsonl

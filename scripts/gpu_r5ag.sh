#!/bin/bash
# SpSpMM stress: ticket chunks -- count kernel 8 (shipped) / 16, accumulation 2 (shipped) / 1 / 4
mkdir -p gpurun_out/r5ag
R=$PWD
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ag/ab.jsonl; }
for rep in 1 2; do
  run shipped_cnt8_acc2
  run acc1 LD_PRELOAD=$R/build/ab/acc1.so
  run acc4 LD_PRELOAD=$R/build/ab/acc4.so
  run cnt16 LD_PRELOAD=$R/build/ab/cnt16.so
done
cat gpurun_out/r5ag/ab.jsonl

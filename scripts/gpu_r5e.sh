#!/bin/bash
mkdir -p gpurun_out/r5e
for v in shipped wave0 r4 batch16 batch4 shipped; do
  VARIANT=$v LD_PRELOAD=$PWD/build/ab/$v.so timeout 120 python scripts/ab_spspmm_r5.py stress c4 2>/dev/null | tee -a gpurun_out/r5e/ab.jsonl
done
CHECK=1 timeout 200 python scripts/ab_spspmm_r5.py stress 2>/dev/null | tee -a gpurun_out/r5e/ab.jsonl

#!/bin/bash
# SpSpMM stress: bin tickets drawn 8 (shipped) / 1 / 32 at a time
mkdir -p gpurun_out/r5af
R=$PWD
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5af/ab.jsonl; }
for rep in 1 2; do
  run shipped_chunk8
  run chunk1 LD_PRELOAD=$R/build/ab/chunk1.so
  run chunk32 LD_PRELOAD=$R/build/ab/chunk32.so
done
CHECK=1 VARIANT=shipped_check python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5af/ab.jsonl
cat gpurun_out/r5af/ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5af/prof -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5af/prof.log 2>&1
f=$(find $R/gpurun_out/r5af/prof -name '*kernel_stats.csv' | head -1); python $R/scripts/kstats.py $f --tsamd | head -9
cd $R; python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py -q -m gpu -k "spspmm" 2>&1 | tail -2

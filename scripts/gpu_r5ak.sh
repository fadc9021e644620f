#!/bin/bash
mkdir -p gpurun_out/r5ak
R=$PWD
python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py tests/test_random_cases_gpu.py -q -m gpu -k "spspmm or random" 2>&1 | tail -2
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ak/ab.jsonl; }
for rep in 1 2 3; do
  run shipped_batched
  run old LD_PRELOAD=$R/build/ab/oldsmall.so
done
cat gpurun_out/r5ak/ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5ak/prof -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5ak/prof.log 2>&1
f=$(find $R/gpurun_out/r5ak/prof -name '*kernel_stats.csv' | head -1); python $R/scripts/kstats.py $f --tsamd | grep "smallbin" | head

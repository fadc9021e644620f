#!/bin/bash
mkdir -p gpurun_out/r5ai
R=$PWD
run() { name=$1; shift; env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5ai/ab.jsonl; }
for rep in 1 2 3; do
  run shipped_blockagg
  run oldplan LD_PRELOAD=$R/build/ab/oldplan.so
done
cat gpurun_out/r5ai/ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5ai/prof -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5ai/prof.log 2>&1
f=$(find $R/gpurun_out/r5ai/prof -name '*kernel_stats.csv' | head -1); python $R/scripts/kstats.py $f --tsamd | grep "bin_kernel\|count_kernel" | head

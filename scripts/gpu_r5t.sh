#!/bin/bash
# SpSpMM stress: occupancy of the hist / bin kernels (dynamic LDS counters) and loads in flight of the one-wave accumulation
mkdir -p gpurun_out/r5t
R=$PWD
run() { # name, env...
  name=$1; shift
  env VARIANT=$name "$@" python scripts/ab_spspmm_r5.py stress c4 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5t/ab.jsonl
}
for rep in 1 2; do
  run shipped_dynlds
  run static_counters TSAMD_SPSPMM_STATIC_COUNTERS=1
  run wb16 LD_PRELOAD=$R/build/ab/wb16.so
  run wb32 LD_PRELOAD=$R/build/ab/wb32.so
done
CHECK=1 VARIANT=shipped_check python scripts/ab_spspmm_r5.py stress 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r5t/ab.jsonl
cat gpurun_out/r5t/ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5t/prof -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5t/prof.log 2>&1
f=$(find $R/gpurun_out/r5t/prof -name '*kernel_stats.csv' | head -1); python $R/scripts/kstats.py $f --tsamd | head -16

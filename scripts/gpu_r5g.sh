#!/bin/bash
mkdir -p gpurun_out/r5g
( time timeout 300 python -m pytest tests/test_api_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "spspmm" ) 2>&1 | tail -6
for v in shipped nopairs shipped; do
  VARIANT=$v LD_PRELOAD=$PWD/build/ab/$v.so timeout 120 python scripts/ab_spspmm_r5.py stress c4 2>/dev/null | tee -a gpurun_out/r5g/ab.jsonl
done
VARIANT=shipped_sub0 TSAMD_SPSPMM_SUBBINS=0 timeout 120 python scripts/ab_spspmm_r5.py stress 2>/dev/null | tee -a gpurun_out/r5g/ab.jsonl
VARIANT=nopairs_sub0 TSAMD_SPSPMM_SUBBINS=0 LD_PRELOAD=$PWD/build/ab/nopairs.so timeout 120 python scripts/ab_spspmm_r5.py stress 2>/dev/null | tee -a gpurun_out/r5g/ab.jsonl
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5g/prof -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5g/prof.log 2>&1
f=$(find $R/gpurun_out/r5g/prof -name '*kernel_stats.csv' | head -1)
python $R/scripts/kstats.py $f --tsamd | head -12

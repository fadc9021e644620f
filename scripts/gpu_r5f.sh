#!/bin/bash
mkdir -p gpurun_out/r5f
R=$PWD
cd /tmp && export TMPDIR=/tmp
for sb in 1 0; do
  TSAMD_SPSPMM_SUBBINS=$sb timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5f/prof_sub$sb -o stress -- python $R/scripts/ab_spspmm_r5.py stress > $R/gpurun_out/r5f/prof_sub$sb.log 2>&1
  f=$(find $R/gpurun_out/r5f/prof_sub$sb -name '*kernel_stats.csv' | head -1)
  echo "== subbins=$sb"; tail -1 $R/gpurun_out/r5f/prof_sub$sb.log; python $R/scripts/kstats.py $f --tsamd | head -22
done

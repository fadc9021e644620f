#!/bin/bash
# kernel timeline of the functional coalesce, fused reduction
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6ac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o co -- python scripts/prof_sort.py coalesce > $OUT/prof.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"; tail -3 $OUT/prof.log
python scripts/kstats.py $f | head -30 | tee $OUT/coalesce_kernel_stats.txt
rm -f $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv

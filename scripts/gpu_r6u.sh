#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6u
TSAMD_LIB=build/ab/fwrec.so python scripts/ab_fwd_winrec.py 2>&1 | tee gpurun_out/r6u/ab_fwd_winrec.jsonl | cut -c1-400
cd /tmp && export TMPDIR=/tmp
TSAMD_LIB=$GRAFT_REPO_ROOT/build/ab/fwrec.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6u/prof -o fw -- python $GRAFT_REPO_ROOT/scripts/ab_fwd_winrec.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r6u/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} | cut -c1-220

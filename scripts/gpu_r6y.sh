#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6y
rm -f gpurun_out/r6y/ab_sum_snap.jsonl
for rep in 1 2; do for v in base snapall64 snapall128; do
TSAMD_LIB=build/ab/$v.so python scripts/ab_sum_snap.py 2>/dev/null | tee -a gpurun_out/r6y/ab_sum_snap.jsonl | cut -c1-400
done; done

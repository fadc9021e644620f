#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6y
for rep in 1 2; do for v in base snap64 snap128; do
TSAMD_LIB=build/ab/$v.so python scripts/ab_minmax_snap.py 2>/dev/null | tee -a gpurun_out/r6y/ab_minmax_snap.jsonl | cut -c1-500
done; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6y
for rep in 1 2; do for v in bl0 bl1; do
echo "== $v"
ONLY_FIRST=1 TSAMD_LIB=build/ab/$v.so python scripts/ab_fwd_winrec.py 2>/dev/null | cut -c1-330
done; done

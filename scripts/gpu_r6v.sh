#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_records_gpu.py -m gpu -x -q 2>&1 | tail -4
python scripts/ab_fwd_winrec.py 2>&1 | tee gpurun_out/r6v/ab_fwd_winrec.jsonl | cut -c1-700
WIDE=1 python scripts/ab_fwd_winrec.py 2>&1 | tee gpurun_out/r6v/ab_fwd_winrec_wide.jsonl | cut -c1-700

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6y
timeout 900 python -m pytest tests/test_records_gpu.py -m gpu -x -q 2>&1 | tail -3
python scripts/ab_fwd_winrec.py 2>/dev/null | tee gpurun_out/r6y/ab_fwd_winrec.jsonl | cut -c1-400
WIDE=1 python scripts/ab_fwd_winrec.py 2>/dev/null | tee gpurun_out/r6y/ab_fwd_winrec_wide.jsonl | cut -c1-400

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6v
timeout 600 python -m pytest tests/test_records_gpu.py -m gpu -x -q 2>&1 | tail -12



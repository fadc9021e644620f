#!/bin/bash
mkdir -p gpurun_out/r5c
timeout 200 python scripts/profile_slow_tests.py > gpurun_out/r5c/profile.log 2>&1
grep -v "^$" gpurun_out/r5c/profile.log | grep -v "Ordered by\|List reduced\|function calls" | cut -c1-170 | head -90
( time timeout 150 python -m pytest tests/test_parallel_gpu.py tests/test_spmm_coo_gpu.py -m gpu -x -q ) 2>&1 | tail -8
python - <<'PY' 2>&1 | tail -5
import sys, torch
sys.path.insert(0, '.')
from tests import baseline_configs as bc
import pytorch_sparse_amd
r = bc.run_c1(torch.device('cuda:0'), cpu=True)
print({k: r[k] for k in ('ms', 'ms_device', 'gedges_per_s')}, r['parity']['ok'], r['cpu_baseline']['ms'])
PY

#!/bin/bash
mkdir -p gpurun_out/r5b
python scripts/time_host_pieces.py > gpurun_out/r5b/host_pieces.log 2>&1
cat gpurun_out/r5b/host_pieces.log
( time python -m pytest tests/test_spmm_coo_gpu.py tests/test_parallel_gpu.py tests/test_sort_gpu.py -m gpu -x -q --durations=8 ) > gpurun_out/r5b/pytest.log 2>&1
tail -30 gpurun_out/r5b/pytest.log
python - <<'PY' 2>&1 | tail -20
import sys, torch
sys.path.insert(0, '.')
from tests import baseline_configs as bc
import pytorch_sparse_amd
r = bc.run_c1(torch.device('cuda:0'), cpu=True)
print({k: r[k] for k in ('ms', 'ms_device', 'gedges_per_s')}, r['parity']['ok'], r['cpu_baseline']['ms'])
PY

#!/bin/bash
mkdir -p gpurun_out/r5d
( time timeout 300 python -m pytest tests/test_api_gpu.py -m gpu -x -q -k "spspmm" ) 2>&1 | tail -8
timeout 400 python - <<'PY' 2>&1 | tail -12
import sys, json, torch
sys.path.insert(0, '.')
from tests import baseline_configs as bc
import pytorch_sparse_amd
dev = torch.device('cuda:0')
torch.set_num_threads(32)
r = bc.run_spspmm(dev, 'stress', cpu=False, iters=3)
print(json.dumps({k: r[k] for k in ('ms', 'gproducts_per_s')}), r['parity'])
r = bc.run_spspmm(dev, 'c4', cpu=False, iters=5)
print('c4', r['ms'])
PY
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r5d/prof -o stress -- python /root/repo/scripts/prof_spspmm.py stress > /root/repo/gpurun_out/r5d/prof.log 2>&1
f=$(find gpurun_out/r5d/prof -name '*kernel_stats.csv' | head -1); python scripts/kstats.py $f --tsamd | head -30

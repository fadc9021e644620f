"""Experiment: partition boundaries snapped to row starts (TSAMD_EXP_MINMAX_SNAP, see spmm_partition_kernel) for every
min / max forward, not only the record-writing one: configs[2] forward through the C-ABI, int64 and int32 winners.
Run once per variant library (TSAMD_LIB).  -> one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_sparse_amd import _native as nat  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from tests.baseline_configs import gpu_ms_stream, rmat_graph  # noqa: E402

dev = torch.device('cuda:0')
rp, c, n = rmat_graph(20, 20, dev)
E = c.numel()
res = dict(lib=os.path.basename(os.environ.get('TSAMD_LIB', 'shipped')))
for dtype, K in ((torch.bfloat16, 128), (torch.float32, 128), (torch.bfloat16, 64)):
    x = synth.features(n, K, dtype=dtype, device=dev)
    key = '%s_%d' % (str(dtype).split('.')[1], K)
    t64 = [round(gpu_ms_stream(lambda: nat.spmm(rp, c, None, x, 'max'), iters=20), 4) for _ in range(3)]
    t32 = [round(gpu_ms_stream(lambda: nat.spmm_minmax_arg32(rp, c, None, x, 'max'), iters=20), 4) for _ in range(3)]
    res[key] = dict(arg64=min(t64), arg32=min(t32))
    o, a = nat.spmm(rp, c, None, x, 'max')
    res[key]['checksum'] = [int(a.sum()), float(o.float().sum())]
print(json.dumps(res), flush=True)

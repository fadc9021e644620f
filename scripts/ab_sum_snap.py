"""Experiment: snapped partition boundaries for the SUM forward (north star 2^21 R-MAT F = 128 fp32, configs[1] F = 64,
configs[4] share F = 256) through the C-ABI.  Run once per variant library (TSAMD_LIB).  -> one JSON line."""
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
res = dict(lib=os.path.basename(os.environ.get('TSAMD_LIB', 'shipped')))
for scale, K in ((21, 128), (20, 64), (21, 256)):
    rp, c, n = rmat_graph(scale, 20, dev)
    E = c.numel()
    v = synth.values(E, device=dev)
    x = synth.features(n, K, device=dev)
    t = [round(gpu_ms_stream(lambda: nat.spmm(rp, c, v, x, 'sum'), iters=30), 4) for _ in range(4)]
    o, _ = nat.spmm(rp, c, v, x, 'sum')
    res['s%d_F%d' % (scale, K)] = dict(ms=t, checksum=float(o.double().sum()))
    del rp, c, v, x, o
    torch.cuda.empty_cache()
print(json.dumps(res), flush=True)

"""tsamd_spmm_minmax_bw_csc on int64 ids against int32 ids (same values), configs[2] graph, bf16 F = 128 -> JSON."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import _native as nat  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from tests.baseline_configs import gpu_ms, rmat_graph  # noqa: E402

dev = torch.device('cuda:0')
rp, c, n = rmat_graph(20, 20, dev)
E, K = c.numel(), 128
A = ts.SparseTensor(rowptr=rp, col=c, value=None, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
colptr, perm, row = A.storage.colptr(), A.storage.csr2csc(), A.storage.row()
for dtype in (torch.bfloat16, torch.float32):
    x = synth.features(n, K, dtype=dtype, device=dev)
    g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
    for has_value in (False, True):
        v = synth.values(E, dtype=dtype, device=dev) if has_value else None
        out, arg = nat.spmm(rp, c, v, x, 'max')
        arg32 = arg.int()
        res = dict(dtype=str(dtype).split('.')[1], has_value=has_value)
        for rep in range(2):
            for name, a in (('i64', arg), ('i32', arg32)):
                res['%s_%d_ms' % (name, rep)] = round(gpu_ms(lambda: nat.spmm_minmax_bw_csc(rp, c, v, x, g, a, colptr, perm, row, want_value=has_value, want_mat=True), iters=10), 4)
        print(json.dumps(res), flush=True)

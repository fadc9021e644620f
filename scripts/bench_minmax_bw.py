"""configs[2] backward (2^20 R-MAT, F = 128 bf16, max): the three grad_mat routes (+ grad_value) -> JSON lines.
TSAMD_MINMAX_BW_LISTS=0 selects the round-3 pull (win masks + masked merge-path SpMM)."""
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
dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[os.environ.get('DTYPE', 'bf16')]
rp, c, n = rmat_graph(20, 20, dev)
E, K = c.numel(), int(os.environ.get('K', 128))
x = synth.features(n, K, dtype=dtype, device=dev)
g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
A = ts.SparseTensor(rowptr=rp, col=c, value=None, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
colptr, perm, row = A.storage.colptr(), A.storage.csr2csc(), A.storage.row()
for has_value in (False, True):
    v = synth.values(E, dtype=dtype, device=dev) if has_value else None
    out, arg = torch.ops.torch_sparse.spmm_max(rp, c, v, x)
    res = dict(has_value=has_value, K=K, E=E)
    for name, env in (('lists', '1'), ('masks', None)):
        if env is not None:
            os.environ['TSAMD_MINMAX_BW_LISTS'] = env
        else:
            os.environ.pop('TSAMD_MINMAX_BW_LISTS', None)
        res[name + '_mat_ms'] = round(gpu_ms(lambda: nat.spmm_minmax_bw_csc(rp, c, v, x, g, arg, colptr, perm, row, want_value=False, want_mat=True), iters=8), 4)
        if has_value:
            res[name + '_mat_value_ms'] = round(gpu_ms(lambda: nat.spmm_minmax_bw_csc(rp, c, v, x, g, arg, colptr, perm, row, want_value=True, want_mat=True), iters=8), 4)
        gm = nat.spmm_minmax_bw_csc(rp, c, v, x, g, arg, colptr, perm, row, want_value=False, want_mat=True)[1]
        res[name + '_sum'] = float(gm.double().sum())
        if name == 'lists':
            ref = gm
        else:
            res['lists_equal_masks_bits'] = bool(torch.equal(gm, ref))
    os.environ.pop('TSAMD_MINMAX_BW_LISTS', None)
    res['scatter_mat_ms'] = round(gpu_ms(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=False, want_mat=True), iters=8), 4)
    if has_value:
        res['scatter_mat_value_ms'] = round(gpu_ms(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=True, want_mat=True), iters=8), 4)
        res['value_only_ms'] = round(gpu_ms(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=True, want_mat=False), iters=8), 4)
    print(json.dumps(res), flush=True)

"""VERDICT r5 item 4 measured: configs[2] (2^20 R-MAT, F = 128, max) forward + pull backward with the winners kept as
int32 ids (tsamd_spmm_minmax_arg32 + tsamd_spmm_minmax_bw_csc_arg32: round 5) against the forward that leaves the
backward's winner records (tsamd_spmm_minmax_records + tsamd_spmm_minmax_bw_csc_records: round 6), C-ABI level, same
box, gradients compared bit for bit.  -> JSON lines."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_sparse_amd import _native as nat  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from tests.baseline_configs import gpu_ms_stream, rmat_graph  # noqa: E402
import pytorch_sparse_amd as ts  # noqa: E402

dev = torch.device('cuda:0')
rp, c, n = rmat_graph(20, 20, dev)
E = c.numel()
A = ts.SparseTensor(rowptr=rp, col=c, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
colptr, perm, row = A.storage.colptr(), A.storage.csr2csc(), A.storage.row()
CASES = ((torch.bfloat16, 128), (torch.float32, 128), (torch.float16, 100))
if os.environ.get("WIDE"):
    CASES = ((torch.bfloat16, 256), (torch.bfloat16, 64), (torch.float32, 64), (torch.float32, 256))
if os.environ.get("ONLY_FIRST"):
    CASES = CASES[:1]
for dtype, K in CASES:
    for has_value in (False, True):
        x = synth.features(n, K, dtype=dtype, device=dev)
        g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
        v = synth.values(E, dtype=dtype, device=dev) if has_value else None
        wv = has_value and (K * x.element_size()) % 16 == 0
        res = dict(dtype=str(dtype).split('.')[1], K=K, has_value=has_value)
        for rep in range(2):
            fw_a = lambda: nat.spmm_minmax_arg32(rp, c, v, x, 'max')  # noqa: E731
            fw_r = lambda: nat.spmm_minmax_records(rp, c, v, x, 'max', row)  # noqa: E731
            out_a, arg = fw_a()
            out_r, rec = fw_r()
            bw_a = lambda: nat.spmm_minmax_bw_csc(rp, c, v, x, g, arg, colptr, perm, row, want_value=wv, want_mat=True)  # noqa: E731
            bw_r = lambda: nat.spmm_minmax_bw_csc_records(rp, c, has_value, x, g, rec, colptr, perm, row, want_value=wv)  # noqa: E731
            for name, fn in (('ids_fw', fw_a), ('rec_fw', fw_r), ('ids_bw', bw_a), ('rec_bw', bw_r)):
                res.setdefault(name + '_ms', []).append(round(gpu_ms_stream(fn, iters=20), 4))
        ga, gr = bw_a(), bw_r()
        res['same_out'] = bool(torch.equal(out_a, out_r))
        res['same_grad_mat'] = bool(torch.equal(ga[1], gr[1]))
        res['same_grad_value'] = None if not wv else bool(torch.equal(ga[0], gr[0]))
        res['ids_step_ms'] = round(min(res['ids_fw_ms']) + min(res['ids_bw_ms']), 4)
        res['rec_step_ms'] = round(min(res['rec_fw_ms']) + min(res['rec_bw_ms']), 4)
        print(json.dumps(res), flush=True)
        del x, g, v, out_a, out_r, arg, rec, ga, gr

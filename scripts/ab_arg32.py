"""configs[2] (2^20 R-MAT, F = 128 bf16 / fp32, max) through SparseTensor.matmul: forward + backward with the winners kept
as int32 ids (default since round 5) against int64 ids (the spmm_max op, want_arg) -> JSON lines."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from pytorch_sparse_amd.tensor import storage_spmm  # noqa: E402
from tests.baseline_configs import gpu_ms, rmat_graph  # noqa: E402

dev = torch.device('cuda:0')
rp, c, n = rmat_graph(20, 20, dev)
E = c.numel()
for dtype, K in ((torch.bfloat16, 128), (torch.float32, 128), (torch.bfloat16, 64)):
    for has_value in (False, True):
        x = synth.features(n, K, dtype=dtype, device=dev).requires_grad_()
        g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
        v = synth.values(E, dtype=dtype, device=dev).requires_grad_() if has_value else None
        A = ts.SparseTensor(rowptr=rp, col=c, value=v, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
        A.storage.colptr(), A.storage.csr2csc(), A.storage.row()
        res = dict(dtype=str(dtype).split('.')[1], K=K, has_value=has_value)
        grads = {}
        for name, want_arg in (('arg32', False), ('arg64', True), ('arg32_b', False), ('arg64_b', True)):
            def fw():
                return storage_spmm(A.storage, x, 'max', want_arg)

            def fwbw():
                x.grad = None
                if v is not None:
                    v.grad = None
                out, arg = storage_spmm(A.storage, x, 'max', want_arg)
                out.backward(g)

            out, arg = fw()
            res[name + '_argdtype'] = str(arg.dtype).split('.')[1]
            res[name + '_fw_ms'] = round(gpu_ms(fw, iters=10), 4)
            res[name + '_fwbw_ms'] = round(gpu_ms(fwbw, iters=10), 4)
            fwbw()
            grads[name] = (x.grad.clone(), None if v is None else v.grad.clone())
        res['same_bits'] = bool(torch.equal(grads['arg32'][0], grads['arg64'][0]) and
                                (v is None or torch.equal(grads['arg32'][1], grads['arg64'][1])))
        print(json.dumps(res), flush=True)

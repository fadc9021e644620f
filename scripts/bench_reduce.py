"""Row / column reductions of a SparseTensor on the north-star graph.  One JSON object per line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')


def gpu_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        t.append(s.elapsed_time(e))
    t.sort()
    return t[len(t) // 2]


rp, c = synth.rmat_csr(21, 20, seed=0, device=dev)
n = 1 << 21
E = c.numel()
A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(E, device=dev), sparse_sizes=(n, n), is_sorted=True,
                    trust_data=True)
A.storage.fill_cache_()
row = A.storage.row()
val = A.storage.value()
for reduce in ('sum', 'mean', 'max'):
    for dim in (1, 0):
        ms = gpu_ms(lambda: getattr(A, reduce)(dim=dim))
        index = row if dim == 1 else c
        red = {'sum': 'sum', 'mean': 'mean', 'max': 'amax'}[reduce]
        ref = gpu_ms(lambda: torch.zeros(n, device=dev).scatter_reduce(0, index, val, red, include_self=False))
        print(json.dumps(dict(bench='reduce', reduce=reduce, dim=dim, E=E, ms=round(ms, 3),
                              torch_scatter_reduce_ms=round(ref, 3), gbs=round(E * 4 / ms / 1e6, 1))), flush=True)
A2 = A.set_value(torch.stack([val, val, val], 1).contiguous(), layout='coo')
print(json.dumps(dict(bench='reduce', reduce='sum', dim=1, D=3, ms=round(gpu_ms(lambda: A2.sum(dim=1)), 3))), flush=True)

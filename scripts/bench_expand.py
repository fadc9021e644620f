"""ptr2ind / select_segments timing on skewed row-length laws (the kernels that expand a pointer array
into per-entry ids).  Prints one JSON object per line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd  # noqa: E402,F401
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')


def gpu_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def ptr_from_deg(deg):
    p = torch.zeros(deg.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=p[1:])
    return p


cases = {}
rp, c = synth.rmat_csr(21, 20, seed=0, device=dev)
cases['rmat21'] = rp
M = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
cases['uniform20'] = ptr_from_deg(torch.full((M, ), 20, dtype=torch.int64, device=dev))
d = torch.randint(0, 8, (M, ), generator=g, device=dev)
d[12345] = 30_000_000
cases['one_hub_30M'] = ptr_from_deg(d)
d = torch.zeros(M, dtype=torch.int64, device=dev)
d[torch.randint(0, M, (2000, ), generator=g, device=dev)] = 20_000
cases['2000_rows_of_20k'] = ptr_from_deg(d)
cases['deg1'] = ptr_from_deg(torch.ones(1 << 25, dtype=torch.int64, device=dev))

for name, p in cases.items():
    E = int(p[-1])
    ms = gpu_ms(lambda: torch.ops.torch_sparse.ptr2ind(p, E))
    print(json.dumps(dict(bench='ptr2ind', case=name, rows=p.numel() - 1, E=E, ms=round(ms, 3),
                          gbs=round(E * 8 / ms / 1e6, 1))), flush=True)
    ind = torch.arange(E, device=dev)
    idx = torch.arange(p.numel() - 1, device=dev)
    ms = gpu_ms(lambda: torch.ops.tsamd.select_segments(p, ind, idx, True, True))
    print(json.dumps(dict(bench='select_all_segments', case=name, E=E, ms=round(ms, 3),
                          gbs=round(E * 32 / ms / 1e6, 1))), flush=True)

# ind2ptr: sorted row ids -> row pointer, incl. a matrix whose entries all sit in the last of 2^25 rows
row21 = torch.ops.torch_sparse.ptr2ind(cases['rmat21'], int(cases['rmat21'][-1]))
ind_cases = {'rmat21': (row21, 1 << 21),
             'all_in_last_row_of_2^25': (torch.full((1 << 20, ), (1 << 25) - 1, dtype=torch.int64, device=dev), 1 << 25),
             'every_1000th_row': (torch.arange(1 << 20, device=dev) * 1000, (1 << 20) * 1000),
             'few_rows_many_entries': (torch.arange(1 << 25, device=dev) // (1 << 15), 1 << 10)}
for name, (ind, Mrows) in ind_cases.items():
    ms = gpu_ms(lambda: torch.ops.torch_sparse.ind2ptr(ind, Mrows))
    print(json.dumps(dict(bench='ind2ptr', case=name, rows=Mrows, E=ind.numel(), ms=round(ms, 3))), flush=True)

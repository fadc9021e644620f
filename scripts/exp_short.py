"""How much of the north-star time is per-row overhead of short / empty rows?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
rowptr, col = synth.rmat_csr(21, 20, seed=0, device=dev); n = 1 << 21
E = col.numel(); val = synth.values(E, device=dev); x = synth.features(n, 128, device=dev)
deg = rowptr[1:] - rowptr[:-1]; row = nat.ptr2ind(rowptr, E)
def timeit(rp, c, v, tag):
    for _ in range(3): nat.spmm(rp, c, v, x, 'sum')
    ts = []
    for _ in range(9):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rp, c, v, x, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); print('%-46s rows %8d edges %9d  %.3f ms' % (tag, rp.numel() - 1, c.numel(), ts[4]), flush=True)
timeit(rowptr, col, val, 'full')
keep_row = deg >= 8
keep = keep_row[row]
d2 = torch.where(keep_row, deg, torch.zeros_like(deg)); rp2 = torch.zeros_like(rowptr); torch.cumsum(d2, 0, out=rp2[1:])
timeit(rp2, col[keep].contiguous(), val[keep].contiguous(), 'rows with deg<8 emptied (same M)')
# drop empty rows entirely (compact row space): removes the per-empty-row cost
nz = d2 > 0
rp3 = torch.zeros(int(nz.sum()) + 1, dtype=torch.int64, device=dev); torch.cumsum(d2[nz], 0, out=rp3[1:])
timeit(rp3, col[keep].contiguous(), val[keep].contiguous(), 'only rows with deg>=8, no empty rows')
nz0 = deg > 0
rp4 = torch.zeros(int(nz0.sum()) + 1, dtype=torch.int64, device=dev); torch.cumsum(deg[nz0], 0, out=rp4[1:])
timeit(rp4, col, val, 'all edges, empty rows removed')

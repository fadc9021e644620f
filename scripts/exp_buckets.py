"""Where does R-MAT time go?  Keep only rows whose degree falls in a bucket (others emptied)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
scale, K = 21, 128
rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
deg = rowptr[1:] - rowptr[:-1]
val = synth.values(col.numel(), device=dev); x = synth.features(n, K, device=dev)
row = nat.ptr2ind(rowptr, col.numel())
def timeit(rp, c, v):
    for _ in range(2): nat.spmm(rp, c, v, x, 'sum')
    ts = []
    for _ in range(7):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rp, c, v, x, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts) // 2]
print('full: %.3f ms  E=%d' % (timeit(rowptr, col, val), col.numel()))
for lo, hi in ((0, 1), (1, 8), (8, 64), (64, 512), (512, 1 << 30), (1, 513), (0, 513)):
    keep_row = (deg >= lo) & (deg < hi)
    keep = keep_row[row]
    c2 = col[keep].contiguous(); v2 = val[keep].contiguous()
    d2 = torch.where(keep_row, deg, torch.zeros_like(deg))
    rp2 = torch.zeros_like(rowptr); torch.cumsum(d2, 0, out=rp2[1:])
    if c2.numel() == 0:
        c2 = torch.zeros(1, dtype=torch.int64, device=dev); v2 = torch.zeros(1, device=dev)
        E2 = 0
    else:
        E2 = c2.numel()
    t = timeit(rp2, c2, v2)
    print('deg [%d,%d): rows %d edges %d  %.3f ms  %.2f GE/s' % (lo, hi, int(keep_row.sum()), E2, t, E2 / t / 1e6), flush=True)
# same buckets but with rows compacted to the front (no empty rows in between)

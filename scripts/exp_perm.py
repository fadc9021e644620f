"""Does the placement of hot X rows matter?  R-MAT with column ids relabelled by a random permutation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
scale, K = 21, 128
rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
E = col.numel()
val = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
def timeit(rp, c, tag):
    for _ in range(3): nat.spmm(rp, c, val, x, 'sum')
    ts = []
    for _ in range(9):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rp, c, val, x, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); t = ts[len(ts) // 2]
    print('%-50s %.3f ms %.2f GE/s' % (tag, t, c.numel() / t / 1e6), flush=True)
timeit(rowptr, col, 'rmat')
g = torch.Generator(device=dev); g.manual_seed(1)
perm = torch.randperm(n, generator=g, device=dev)
timeit(rowptr, perm[col], 'rmat, column ids randomly relabelled')
# also rows relabelled (degree no longer correlated with index) -- needs a re-sort: skip, merge-path is order independent
deg = rowptr[1:] - rowptr[:-1]
for rep in range(2):
    timeit(rowptr, col, 'rmat (repeat)')

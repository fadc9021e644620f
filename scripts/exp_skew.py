"""Separate row-degree skew from column-popularity skew (north-star shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
scale, K = 21, int(sys.argv[1]) if len(sys.argv) > 1 else 128
rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
E = col.numel()
val = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
def timeit(rp, c, v, tag):
    for _ in range(2): nat.spmm(rp, c, v, x, 'sum')
    ts = []
    for _ in range(7):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rp, c, v, x, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); t = ts[len(ts) // 2]
    print('%-40s %.3f ms %.2f GE/s' % (tag, t, c.numel() / t / 1e6), flush=True)
timeit(rowptr, col, val, 'rmat rows + rmat cols')
g = torch.Generator(device=dev); g.manual_seed(5)
ucol = torch.randint(0, n, (E,), generator=g, device=dev)
timeit(rowptr, ucol, val, 'rmat rows + uniform cols')
d = E // n
urp = torch.arange(0, (n + 1) * d, d, dtype=torch.int64, device=dev)
perm = torch.randperm(E, generator=g, device=dev)
scol = col[perm][: n * d].contiguous()
timeit(urp, scol, val[: n * d].contiguous(), 'uniform rows(deg %d) + rmat cols (shuffled)' % d)
timeit(urp, col[: n * d].contiguous(), val[: n * d].contiguous(), 'uniform rows + rmat cols (in order)')
timeit(urp, ucol[: n * d].contiguous(), val[: n * d].contiguous(), 'uniform rows + uniform cols')
# rows sorted by degree descending (relabel rows): balanced dispatch?
deg = rowptr[1:] - rowptr[:-1]

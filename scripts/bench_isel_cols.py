"""index_select along the columns: general path (CSC view + re-sort) vs sorted-subset fast path."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth
dev = torch.device('cuda:0')
def wall(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    t=[]
    for _ in range(iters):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter()-t0)
    t.sort(); return t[len(t)//2]*1e3
rp, c = synth.rmat_csr(20, 20, seed=0, device=dev); n = 1<<20; E = c.numel()
A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(E, device=dev), sparse_sizes=(n,n), is_sorted=True, trust_data=True)
g = torch.Generator().manual_seed(0)
idx_rep = torch.randint(0, n, (n//2,), generator=g).to(dev)
idx_uni = torch.randperm(n, generator=g)[:n//2].to(dev)
print(json.dumps(dict(bench='index_select_cols', first_call_ms=round(wall(lambda: ts.SparseTensor(rowptr=rp, col=c, value=A.storage.value(), sparse_sizes=(n,n), is_sorted=True, trust_data=True).index_select(1, idx_uni), iters=3, warm=1),3),
   cached_csc_repeats_ms=round(wall(lambda: A.index_select(1, idx_rep)),3), cached_csc_unique_ms=round(wall(lambda: A.index_select(1, idx_uni)),3),
   sorted_subset_ms=round(wall(lambda: A.index_select(1, idx_uni.sort().values)),3),
   rows_ms=round(wall(lambda: A.index_select(0, idx_uni)),3))))

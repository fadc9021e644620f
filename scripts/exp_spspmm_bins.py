"""Size distribution of the (row, column range) bins of the large-row SpSpMM path on the stress product
(A * A^T, R-MAT scale 19, edge factor 8): cnt[i, q] = sum_{e in A_i} #{entries of B row col(e) in range q},
computed as a sparse x dense product with the package's own SpMM (int64)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth, _native as nat

dev = torch.device('cuda:0')
scale, lg = 19, 13
rp, c = synth.rmat_csr(scale, 8, seed=0, device=dev)
n = 1 << scale
A = ts.SparseTensor(rowptr=rp, col=c, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
At = A.t()
rpB, cB = At.storage.rowptr(), At.storage.col()
rowB = At.storage.row()
nr = n >> lg
L = torch.zeros(n, nr, dtype=torch.int64, device=dev)
L.view(-1).index_add_(0, rowB * nr + (cB >> lg), torch.ones_like(cB))
cnt, _ = nat.spmm(rp, c, None, L, 'sum')          # [n, nr] products per (row, range)
prod = cnt.sum(1)
large = prod > 4096
cl = cnt[large]
flat = cl.flatten()
flat = flat[flat > 0]
srt, _ = torch.sort(flat, descending=True)
tot = int(flat.sum())
res = dict(rows_large=int(large.sum()), bins=int(flat.numel()), products_large=tot, max_bin=int(srt[0]),
           top10=[int(v) for v in srt[:10]],
           share_top={k: round(float(srt[:k].sum()) / tot, 4) for k in (1, 10, 100, 1000, 10000)},
           bins_gt={str(t): int((flat > t).sum()) for t in (512, 4096, 65536, 1 << 20)},
           products_in_bins_gt={str(t): round(float(flat[flat > t].sum()) / tot, 4) for t in (512, 4096, 65536, 1 << 20)})
print(json.dumps(res))

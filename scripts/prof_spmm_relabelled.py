"""Profile target: a few north-star SpMM launches in the relabelled layout (used under rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth
scale, K, iters = 21, 128, 3
dev = torch.device('cuda:0')
rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev)
n = 1 << scale
A = ts.SparseTensor(rowptr=rowptr, col=col, value=synth.values(col.numel(), device=dev), sparse_sizes=(n, n),
                    is_sorted=True, trust_data=True)
x_h = ts.to_relabelled(synth.features(n, K, device=dev))
with torch.no_grad():
    for _ in range(iters):
        ts.matmul_relabelled(A, x_h)
torch.cuda.synchronize()

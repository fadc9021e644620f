import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth
dev = torch.device('cuda:0')
m = n = 500000; nnz = 7500000
row, col = synth.uniform_edges(m, n, nnz, seed=0, device=dev)
A = ts.SparseTensor(row=row, col=col, value=synth.values(nnz, device=dev), sparse_sizes=(m, n)).coalesce()
At = A.t()
for _ in range(3):
    C = A @ At
torch.cuda.synchronize()
print(C.nnz())

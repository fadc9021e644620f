"""Profile target: the R-MAT sort (fallback of the bucket path: one-sweep chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts  # noqa
from pytorch_sparse_amd import synth
dev = torch.device('cuda:0')
row, col = synth.rmat_edges(20, 20, seed=3)
row, col = row.to(dev), col.to(dev)
m = n = 1 << 20
for _ in range(4):
    torch.ops.tsamd.sort_coo(row, col, m, n, True)
torch.cuda.synchronize()

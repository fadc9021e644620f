"""Profile target: a few SpMM launches at a given shape (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reduce = sys.argv[3] if len(sys.argv) > 3 else 'sum'
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev)
n = 1 << scale; E = col.numel()
val = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
for _ in range(iters):
    nat.spmm(rowptr, col, val, x, reduce)
torch.cuda.synchronize()
print('E', E)

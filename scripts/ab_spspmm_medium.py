"""A/B of the medium-row class of SpSpMM (rows of 513..TSAMD_SPSPMM_MEDIUM_CAP products sorted by one 256-thread
workgroup; above that the binned large-row path): stress product (R-MAT) and a uniform product whose rows are all
medium.  Run once per build (the op library is the shipped one)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth
from tests.baseline_configs import gpu_ms
dev = torch.device('cuda:0')
res = {}
rp, c = synth.rmat_csr(19, 8, seed=0, device=dev)
A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(c.numel(), device=dev), sparse_sizes=(1 << 19, 1 << 19), is_sorted=True, trust_data=True)
At = A.t()
res['stress_ms'] = round(gpu_ms(lambda: A @ At, iters=3, warm=2), 3)
del A, At
torch.cuda.empty_cache()
m = 200000
row, col = synth.uniform_edges(m, m, 40 * m, seed=0, device=dev)
A = ts.SparseTensor(row=row, col=col, value=synth.values(row.numel(), device=dev), sparse_sizes=(m, m)).coalesce()
At = A.t()
C = A @ At
res['uniform40_ms'] = round(gpu_ms(lambda: A @ At, iters=3, warm=2), 3)
res['uniform40_nnzC'] = C.nnz()
print(json.dumps(res))

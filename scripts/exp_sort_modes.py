import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth
from tests.baseline_configs import gpu_ms
dev = torch.device('cuda:0'); ops = torch.ops.tsamd
m = n = 500000; E = 7500000
row, col = synth.uniform_edges(m, n, E, seed=0, device=dev); val = synth.values(E, device=dev)
for mode in (0, 1, 3):
    print(mode, 'val', round(gpu_ms(lambda: ops.sort_coo_values(row, col, m, n, mode, None, val), iters=20), 4),
          'noval', round(gpu_ms(lambda: ops.sort_coo_values(row, col, m, n, mode, None, None), iters=20), 4))

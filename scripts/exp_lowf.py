"""F = 8 / 16 / 32 fp32 sum on the north-star graph and on the uniform-degree control, for A/B builds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import synth, _native as nat
from tests.baseline_configs import gpu_ms, b_alg
dev = torch.device('cuda:0')
n = 1 << 21
graphs = {'rmat': synth.rmat_csr(21, 20, seed=0, device=dev), 'uniform': synth.uniform_degree_csr(n, n, 20, seed=5, device=dev)}
out = {}
for gname, (rp, c) in graphs.items():
    E = c.numel(); v = synth.values(E, device=dev)
    for K in (8, 16, 32):
        x = synth.features(n, K, device=dev)
        ms = gpu_ms(lambda: nat.spmm(rp, c, v, x, 'sum'), iters=10)
        out['%s_F%d' % (gname, K)] = round(ms, 4)
print(os.environ.get('TSAMD_LIB', 'default').split('/')[-1], json.dumps(out), flush=True)

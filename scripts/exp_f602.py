import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
rp, c = synth.rmat_csr(18, 20, seed=0, device=dev); n = 1 << 18; E = c.numel()
v = synth.values(E, device=dev)
for K in (602, 600, 50, 41, 47, 128):
    x = synth.features(n, K, device=dev)
    for _ in range(3): nat.spmm(rp, c, v, x, 'sum')
    ts = []
    for _ in range(9):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rp, c, v, x, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); t = ts[4]; balg = E * (12 + K * 4) + n * K * 4
    print('F=%d  %.3f ms  %.2f GE/s  %.2f TB/s' % (K, t, E / t / 1e6, balg / t / 1e9), flush=True)

"""Time tsamd_spmm for one lib variant (TSAMD_LIB) over a set of graphs; prints one line per case."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
tag = os.environ.get('TAG', os.path.basename(os.environ.get('TSAMD_LIB', 'default')))
cases = sys.argv[1:] or ['rmat21:128', 'uni21d20:128', 'uni21d4:128', 'uni21d64:128', 'rmat20:64']
for case in cases:
    g, K = case.split(':'); K = int(K)
    if g.startswith('rmat'):
        scale = int(g[4:]); rowptr, col = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
    else:
        scale, d = g[3:].split('d'); scale = int(scale); n = 1 << scale
        rowptr, col = synth.uniform_degree_csr(n, n, int(d), seed=0, device=dev)
    E = col.numel()
    val = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
    for red in ('sum',):
        for _ in range(3): nat.spmm(rowptr, col, val, x, red)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); nat.spmm(rowptr, col, val, x, red); e.record(); e.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort(); dt = ts[len(ts) // 2] / 1e3
        balg = E * (8 + 4 + K * 4) + (n + 1) * 8 + n * K * 4
        print('%-14s %-12s %s E=%d  %.3f ms  %.2f GE/s  %.2f TB/s (%.1f%%)' % (tag, case, red, E, dt * 1e3, E / dt / 1e9, balg / dt / 1e12, balg / dt / 8e10), flush=True)
    del rowptr, col, val, x

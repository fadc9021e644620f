"""Where do the slow GPU-suite tests spend their time on the GPU box's host?  (one-off diagnosis, round 5)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_sparse_amd import synth  # noqa: E402


def T(label, fn):
    t = time.perf_counter()
    r = fn()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    print('%-60s %8.2f s' % (label, time.perf_counter() - t), flush=True)
    return r


print('threads', torch.get_num_threads(), 'cpus', os.cpu_count())
for nt in (torch.get_num_threads(), 32):
    torch.set_num_threads(nt)
    print('--- torch threads', nt)
    T('rmat_csr(17,16) cpu', lambda: synth.rmat_csr(17, 16, seed=5))
    T('rmat_csr(19,16) cpu', lambda: synth.rmat_csr(19, 16, seed=3))
    row, col = T('rmat_edges(20,4) cpu', lambda: synth.rmat_edges(20, 4, seed=3))
    T('np stable argsort 4M', lambda: np.argsort(row.numpy() * (1 << 20) + col.numpy(), kind='stable'))
    T('torch.unique 8M cpu', lambda: torch.unique(torch.randint(0, 1 << 38, (8000000, ))))
    T('scatter_add_ [4096,1024] f64 cpu', lambda: torch.zeros(4096, 1024, dtype=torch.float64).scatter_add_(
        0, torch.randint(0, 4096, (4096, 1024)), torch.rand(4096, 1024, dtype=torch.float64)))
T('rmat_csr(19,16) cuda', lambda: synth.rmat_csr(19, 16, seed=3, device='cuda'))
T('rmat_csr(19,16) cuda again', lambda: synth.rmat_csr(19, 16, seed=3, device='cuda'))

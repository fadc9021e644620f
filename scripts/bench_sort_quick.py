"""Same-box A/B of the sort family at the configs[3] input (7.5 M draws over 500k x 500k): one JSON line.
LD_PRELOAD=build/ab/<variant>.so python scripts/bench_sort_quick.py [tag] [--big] [--rmat]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from tests.baseline_configs import gpu_ms, wall_ms  # noqa: E402

dev = torch.device('cuda:0')
ops = torch.ops.tsamd
if os.environ.get('TSAMD_COALESCE_UNFUSED'):  # A/B: the unfused functional coalesce / transpose
    import pytorch_sparse_amd.coalesce as _co
    _co._FUSE_REDUCE = False
tag = next((a for a in sys.argv[1:] if not a.startswith('--')), 'base')
cases = [('c4', 500000, 500000, 7500000)]
if '--big' in sys.argv:
    cases.append(('75m', 1 << 22, 1 << 22, 75000000))
if '--mid' in sys.argv:
    cases.append(('21m', 1 << 20, 1 << 20, 21000000))
for (name, m, n, E) in cases:
    row, col = synth.uniform_edges(m, n, E, seed=0, device=dev)
    val = synth.values(E, device=dev)
    index = torch.stack([row, col])
    r = dict(tag=tag, case=name)
    r['sort'] = round(gpu_ms(lambda: ops.sort_coo(row, col, m, n, True), iters=20), 4)
    r['perm_only'] = round(gpu_ms(lambda: ops.sort_coo(col, row, n, m, False), iters=20), 4)
    r['sort_val'] = round(gpu_ms(lambda: ops.sort_coo_values(row, col, m, n, 3, None, val), iters=20), 4)
    # the compacting chain alone (no host sync inside the op): differences between builds are kernel time
    r['coalesce_op'] = round(gpu_ms(lambda: ops.sort_coalesce_reduce(row, col, m, n, val, 0), iters=20), 4)

    def ctor():
        A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n))
        A.storage.rowptr()
        return A
    A = ctor()

    def t_fresh():
        st = A.storage
        st._csr2csc = None
        st._csc2csr = None
        st._colptr = None
        st._colcount = None
        return A.t()
    rs_ = A.storage.row()
    r['ind2ptr'] = round(gpu_ms(lambda: torch.ops.torch_sparse.ind2ptr(rs_, m), iters=50), 4)
    r['construct'] = round(wall_ms(ctor, 9), 4)
    r['coalesce'] = round(wall_ms(lambda: ts.coalesce(index, val, m, n), 9), 4)
    r['transpose'] = round(wall_ms(lambda: ts.transpose(index, val, m, n), 9), 4)
    r['coalesce_index'] = round(wall_ms(lambda: ts.coalesce(index, None, m, n), 9), 4)
    r['t'] = round(wall_ms(t_fresh, 9), 4)
    print(json.dumps(r), flush=True)
    del row, col, val, index, A
    torch.cuda.empty_cache()
if '--rmat' in sys.argv:
    row, col = synth.rmat_edges(20, 20, seed=3)
    row, col = row.to(dev), col.to(dev)
    m = n = 1 << 20
    r = dict(tag=tag, case='rmat20x20', E=row.numel())
    r['sort'] = round(gpu_ms(lambda: ops.sort_coo(row, col, m, n, True), iters=10), 4)
    print(json.dumps(r), flush=True)

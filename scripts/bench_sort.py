"""Timing of the COO sort / compaction building blocks and of construct / coalesce / transpose / t() on the
configs[3] input (7.5 M draws over 500k x 500k) and on a 75 M-entry input (2^22 x 2^22) -> JSON lines."""
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
big = '--big' in sys.argv
for (m, n, E) in ([(500000, 500000, 7500000)] + ([(1 << 22, 1 << 22, 75000000)] if big else [])):
    row, col = synth.uniform_edges(m, n, E, seed=0, device=dev)
    val = synth.values(E, device=dev)
    index = torch.stack([row, col])
    r = dict(E=E, m=m, n=n)
    r['sort_coo_ms'] = round(gpu_ms(lambda: ops.sort_coo(row, col, m, n, True), iters=10), 4)
    r['sort_perm_only_ms'] = round(gpu_ms(lambda: ops.sort_coo(col, row, n, m, False), iters=10), 4)
    r['sort_coo_auto_ms'] = round(gpu_ms(lambda: ops.sort_coo_auto(row, col, m, n), iters=10), 4)
    rs, cs, perm = ops.sort_coo(row, col, m, n, True)
    r['sort_auto_on_sorted_ms'] = round(gpu_ms(lambda: ops.sort_coo_auto(rs, cs, m, n), iters=10), 4)
    r['coalesce_index_ms'] = round(gpu_ms(lambda: ops.coalesce_index(rs, cs), iters=10), 4)
    r['coo_check_ms'] = round(gpu_ms(lambda: ops.coo_check(row, col), iters=10), 4)

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
    r['construct_ms'] = round(wall_ms(ctor, 5), 4)
    r['coalesce_ms'] = round(wall_ms(lambda: ts.coalesce(index, val, m, n), 5), 4)
    r['transpose_ms'] = round(wall_ms(lambda: ts.transpose(index, val, m, n), 5), 4)
    r['t_ms'] = round(wall_ms(t_fresh, 5), 4)
    s = 4
    r['in_out_bytes'] = 2 * E * (16 + s)
    r['construct_frac_of_peak'] = round(2 * E * (16 + s) / r['construct_ms'] / 1e6 / 8000.0, 4)
    print(json.dumps(r), flush=True)
    del row, col, val, index, rs, cs, perm, A
    torch.cuda.empty_cache()

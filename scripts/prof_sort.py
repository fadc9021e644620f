"""Profile target for the sort family: sort_coo / sort_coo_values / construct / coalesce / t() at the configs[3] input
(and with --big the 75 M-entry one), a few calls each.  Run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')
ops = torch.ops.tsamd
reps = int(os.environ.get('REPS', '5'))
which = set(a for a in sys.argv[1:] if not a.startswith('--'))
sizes = [(500000, 500000, 7500000)] + ([(1 << 22, 1 << 22, 75000000)] if '--big' in sys.argv else [])
for (m, n, E) in sizes:
    row, col = synth.uniform_edges(m, n, E, seed=0, device=dev)
    val = synth.values(E, device=dev)
    index = torch.stack([row, col])

    def run(label, fn):
        if which and label not in which:
            return
        for _ in range(reps + 1):
            fn()
        torch.cuda.synchronize()

    run('sort', lambda: ops.sort_coo(row, col, m, n, True))
    run('sort_values', lambda: ops.sort_coo_values(row, col, m, n, 3, None, val))
    run('construct', lambda: ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n)).storage.rowptr())
    run('coalesce', lambda: ts.coalesce(index, val, m, n))

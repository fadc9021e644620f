"""Times the REFERENCE's Python pipelines beside the ATen restatements ("ports") that `bench.py` times on the GPU
box for the C1 and construct / coalesce rows -- in THIS container, where /root/reference exists (it does not travel
to the GPU box, which is why those two rows say `cpu_kind: "port"`).  Same process layout as make_golden.py: the
reference package runs in a subprocess on a scratch copy with its own op library.

    python tests/golden/time_reference_cpu.py  ->  profiles/r05_cpu_port_vs_reference.json

The ports are the functions of tests/baseline_configs.py (run_c1: ref_pipeline; run_construct: ref_construct /
ref_coalesce), restated here call for call so that the file runs without a GPU.
"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

CHILD = r"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
import torch_sparse
from torch_sparse import SparseTensor, coalesce, spmm
threads = int(os.environ['TS_THREADS'])
torch.set_num_threads(threads)
def best(fn, budget):
    t_end, b, n = time.perf_counter() + budget, 1e9, 0
    while n < 3 or time.perf_counter() < t_end:
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0); n += 1
        if n >= 200: break
    return b, n
z = np.load(os.path.join(os.environ['TS_ROOT'], 'tests', 'golden', 'py7_c1_spmm.npz'))
index, value, x = torch.from_numpy(z['index']), torch.from_numpy(z['value']), torch.from_numpy(z['mat'])
m, n = int(z['m']), int(z['n'])
res = {}
torch.set_num_threads(1)
t, k = best(lambda: spmm(index, value, m, n, x), 2.0)
res['c1_reference_ms'] = t * 1e3
torch.set_num_threads(threads)
g = torch.Generator().manual_seed(0)
M = N = 500000; E = 7500000
row = torch.randint(0, M, (E, ), generator=g); col = torch.randint(0, N, (E, ), generator=g)
val = torch.rand(E, generator=g)
def construct():
    A = SparseTensor(row=row, col=col, value=val, sparse_sizes=(M, N))
    A.storage.rowptr()
    return A
t, k = best(construct, 8.0)
res['construct_reference_ms'] = t * 1e3
ind = torch.stack([row, col])
t, k = best(lambda: coalesce(ind, val, M, N), 8.0)
res['coalesce_reference_ms'] = t * 1e3
print('RESULT ' + json.dumps(res))
"""


def port_times(threads):
    import numpy as np
    import torch

    def best(fn, budget):
        t_end, b, n = time.perf_counter() + budget, 1e9, 0
        while n < 3 or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            fn()
            b = min(b, time.perf_counter() - t0)
            n += 1
            if n >= 200:
                break
        return b

    z = np.load(os.path.join(HERE, 'py7_c1_spmm.npz'))
    ic, vc, xc = torch.from_numpy(z['index']), torch.from_numpy(z['value']), torch.from_numpy(z['mat'])
    m, K = int(z['m']), xc.size(1)
    res = {}
    torch.set_num_threads(1)

    def c1():  # tests/baseline_configs.py run_c1: ref_pipeline
        o = xc.index_select(-2, ic[1])
        o = o * vc.unsqueeze(-1)
        idx = ic[0].unsqueeze(-1).expand_as(o)
        return torch.zeros(m, K).scatter_add_(-2, idx, o)
    res['c1_port_ms'] = best(c1, 2.0) * 1e3
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    M = N = 500000
    E = 7500000
    rc = torch.randint(0, M, (E, ), generator=g)
    cc = torch.randint(0, N, (E, ), generator=g)
    vc2 = torch.rand(E, generator=g)

    def construct():  # run_construct: ref_construct
        idx = rc * N + cc
        p = idx.argsort()
        r2, c2, v2 = rc[p], cc[p], vc2[p]
        return r2, c2, v2, torch._convert_indices_from_coo_to_csr(r2, M)

    def coalesce():  # run_construct: ref_coalesce
        idx = rc * N + cc
        p = idx.argsort()
        ids = idx[p]
        mask = torch.ones_like(ids, dtype=torch.bool)
        mask[1:] = ids[1:] > ids[:-1]
        r2, c2 = rc[p][mask], cc[p][mask]
        seg = mask.cumsum(0) - 1
        return r2, c2, torch.zeros(int(r2.numel()), dtype=vc2.dtype).index_add_(0, seg, vc2[p])
    res['construct_port_ms'] = best(construct, 8.0) * 1e3
    res['coalesce_port_ms'] = best(coalesce, 8.0) * 1e3
    return res


def main():
    import make_golden as mg
    threads = min(32, os.cpu_count() or 1)
    scratch, pkg = mg.make_scratch(['coalesce', 'spmm'], mg.SPMM_SRCS)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .coalesce import coalesce\nfrom .spmm import spmm\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_ROOT=ROOT, TS_THREADS=str(threads))
    out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        sys.exit(out.stdout + out.stderr)
    ref = json.loads([l for l in out.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    port = port_times(threads)
    res = dict(host='build container (not the GPU box)', threads=threads, **{k: round(v, 4) for k, v in ref.items()},
               **{k: round(v, 4) for k, v in port.items()})
    for k in ('c1', 'construct', 'coalesce'):
        res[k + '_port_over_reference'] = round(port[k + '_port_ms'] / ref[k + '_reference_ms'], 3)
    res['note'] = ('reference = /root/reference torch_sparse Python (spmm.py, storage.py, coalesce.py) on its compiled CPU '
                   'ops, torch_scatter stood in for by tests/golden/shims; port = the ATen call chains bench.py times on '
                   'the GPU box (tests/baseline_configs.py); best of repeated runs, C1 single-threaded')
    path = os.path.join(ROOT, 'profiles', 'r05_cpu_port_vs_reference.json')
    with open(path, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

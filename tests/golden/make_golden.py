"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF in this
container (it cannot travel to the GPU box; the fixtures can).

  part 1  spmm_*.npz            : torch.ops.ts_ref.spmm_{sum,mean,min,max}, i.e. the reference's
                                  csrc/cpu/spmm_cpu.cpp compiled unmodified (oracle/build_ref.py)
  part 2  py_*.npz              : the reference *Python* package imported from /root/reference
                                  (coalesce / transpose / t() / SparseTensor ctor / spspmm /
                                  legacy spmm / matmul fwd+bwd).  The package needs
                                  `torch.ops.torch_sparse.*` and `torch_scatter`; a scratch copy
                                  of the op library with the reference's own registration names is
                                  built under $TMPDIR, and tests/golden/shims/torch_scatter stands
                                  in for the pip dependency.  Runs in a subprocess so the
                                  `torch_sparse::` names never meet the product's.

  part 3  py3_*.npz             : the reference Python package again, on the shared case list of
                                  tests/golden/cases3.py (narrow / select / index_select /
                                  masked_select / permute / cat / diag / add / mul), with the
                                  reference's csrc/cpu/diag_cpu.cpp added to the op library.

  part 4  py4_*.npz             : the reference's CPU samplers compiled unmodified (rw, sample,
                                  saint, relabel): random walks together with the floats they
                                  drew, and the deterministic cases of the others.

  part 5  py5_*.npz             : multi-hop neighbor_sample (csrc/cpu/neighbor_sample_cpu.cpp), the
                                  take-all cases, on the CSC view of the part-4 graph.

  part 6  py6_random_cases.npz  : 672 randomised small cases of the whole Python surface (cases6.py).

  part 7  py7_c1_spmm.npz       : BASELINE.json configs[0] at its exact size -- the reference's legacy
                                  torch_sparse.spmm(index, value, 1000, 1000, x) on 5 000 UNSORTED
                                  uniform draws (seed 0; duplicates occur), F = 16 fp32 -- inputs and
                                  the reference's output (also in fp64, the well-conditioned yardstick).

  part 8  py8_hetero_*.npz      : hetero_neighbor_sample / hetero_temporal_neighbor_sample
                                  (csrc/cpu/neighbor_sample_cpu.cpp:135-507) on small random heterogeneous graphs
                                  (3 node types, 5 relations): the take-all cases, directed and undirected, and the
                                  temporal sampler with per-type time stamps.

Usage:  python tests/golden/make_golden.py [part1] ... [part8]   (needs /root/reference)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get('TS_REFERENCE', '/root/reference')


def tonp(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def part1():
    from oracle import ref
    from pytorch_sparse_amd import synth
    from tests.util import CODE
    r = ref.ops()
    n_written = 0
    cases = [('rmat', 8, 6), ('uniform', 200, 700)]
    for gname, a, b in cases:
        if gname == 'rmat':
            rp, c = synth.rmat_csr(a, b, seed=7)
            n = m = 1 << a
        else:
            row, col = synth.uniform_edges(a, 300, b, seed=8)
            rp, c = synth.to_csr(row, col, a, 300)
            m, n = a, 300
        E = c.numel()
        for dtype in CODE:
            for reduce in ('sum', 'mean', 'min', 'max'):
                for has_value in (True, False):
                    K = 5 if has_value else 8
                    if dtype.is_floating_point:
                        v = synth.values(E, dtype=dtype)
                        x = synth.features(n, K, dtype=dtype, batch=(2, ) if has_value else ())
                    else:
                        g = torch.Generator().manual_seed(11)
                        v = torch.randint(-4, 5, (E, ), dtype=dtype, generator=g)
                        x = torch.randint(-9, 9, (n, K), dtype=dtype, generator=g)
                    value = v if has_value else None
                    if reduce == 'sum':
                        out, arg = r.spmm_sum(None, rp, c, value, None, None, x), None
                    elif reduce == 'mean':
                        out, arg = r.spmm_mean(None, rp, c, value, None, None, None, x), None
                    elif reduce == 'min':
                        out, arg = r.spmm_min(rp, c, value, x)
                    else:
                        out, arg = r.spmm_max(rp, c, value, x)
                    d = dict(dtype_code=CODE[dtype], reduce=reduce, rowptr=rp.numpy(), col=c.numpy(),
                             mat=tonp(x), out=tonp(out))
                    if has_value:
                        d['value'] = tonp(v)
                    if arg is not None:
                        d['arg_out'] = arg.numpy()
                    name = 'spmm_%s_%s_%s_%s.npz' % (gname, str(dtype).split('.')[1], reduce,
                                                     'val' if has_value else 'noval')
                    np.savez_compressed(os.path.join(HERE, name), **d)
                    n_written += 1
    print('part 1: wrote %d spmm fixtures' % n_written)


PART2 = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
import torch_sparse
from torch_sparse import SparseTensor, coalesce, transpose, spspmm, spmm
from torch_sparse.matmul import matmul
out_dir = os.environ['TS_OUT']
torch.manual_seed(0)
def save(name, **kw):
    np.savez_compressed(os.path.join(out_dir, name), **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})

# --- coalesce / transpose (functional API), values exactly representable so that the
#     unspecified duplicate order of the reference's unstable sort cannot change the sums
for seed, (m, n, nnz) in enumerate([(50, 40, 600), (1, 1, 5), (300, 7, 2000)]):
    g = torch.Generator().manual_seed(seed)
    index = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
    for vname, value in (('i64', torch.randint(-8, 9, (nnz, 2), generator=g)),
                         ('f32', torch.randint(-8, 9, (nnz,), generator=g).float() / 4)):
        for op in ('add', 'mean', 'min', 'max'):
            if op == 'mean' and vname == 'i64':
                continue
            oi, ov = coalesce(index, value, m, n, op=op)
            save('py_coalesce_%d_%s_%s.npz' % (seed, vname, op), index=index, value=value, m=m, n=n, op=op, out_index=oi, out_value=ov)
        ti, tv = transpose(index, value, m, n)
        save('py_transpose_%d_%s.npz' % (seed, vname), index=index, value=value, m=m, n=n, out_index=ti, out_value=tv)
    oi, ov = coalesce(index, None, m, n)
    save('py_coalesce_%d_none.npz' % seed, index=index, m=m, n=n, out_index=oi)

# --- SparseTensor ctor (sort) + caches + t()
g = torch.Generator().manual_seed(10)
m, n, nnz = 37, 53, 400
key = torch.randperm(m * n, generator=g)[:nnz]
row, col = key // n, key % n
val = torch.randn(nnz, generator=g)
A = SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n))
r2, c2, v2 = A.coo()
rowptr = A.storage.rowptr(); colptr = A.storage.colptr(); csr2csc = A.storage.csr2csc(); csc2csr = A.storage.csc2csr()
At = A.t(); tr, tc, tv = At.coo()
save('py_storage.npz', row=row, col=col, value=val, m=m, n=n, s_row=r2, s_col=c2, s_value=v2, rowptr=rowptr,
     colptr=colptr, csr2csc=csr2csc, csc2csr=csc2csr, rowcount=A.storage.rowcount(), colcount=A.storage.colcount(),
     t_row=tr, t_col=tc, t_value=tv)

# --- SpSpMM (torch.sparse.mm behind the reference's spspmm), fp32 + fp64, with/without values
for seed, (m, k, n, nA, nB) in enumerate([(40, 30, 50, 300, 250), (64, 64, 64, 500, 500), (5, 3, 4, 6, 5)]):
    g = torch.Generator().manual_seed(20 + seed)
    kA = torch.randperm(m * k, generator=g)[:nA].sort().values
    kB = torch.randperm(k * n, generator=g)[:nB].sort().values
    iA = torch.stack([kA // k, kA % k]); iB = torch.stack([kB // n, kB % n])
    for dt in (torch.float32, torch.float64):
        vA = torch.randint(-6, 7, (nA,), generator=g).to(dt) / 2
        vB = torch.randint(-6, 7, (nB,), generator=g).to(dt) / 2
        iC, vC = spspmm(iA, vA, iB, vB, m, k, n)
        save('py_spspmm_%d_%s.npz' % (seed, str(dt).split('.')[1]), iA=iA, vA=vA, iB=iB, vB=vB, m=m, k=k, n=n, iC=iC, vC=vC)
    C = matmul(SparseTensor(row=iA[0], col=iA[1], sparse_sizes=(m, k)), SparseTensor(row=iB[0], col=iB[1], sparse_sizes=(k, n)))
    cr, cc, cv = C.coo()
    assert cv is None
    save('py_spspmm_%d_noval.npz' % seed, iA=iA, iB=iB, m=m, k=k, n=n, iC=torch.stack([cr, cc]))

# --- legacy functional spmm (unsorted + duplicate indices allowed) and matmul fwd+bwd
g = torch.Generator().manual_seed(30)
m, n, nnz, F = 20, 25, 150, 6
index = torch.stack([torch.randint(0, m, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)])
value = torch.randn(nnz, generator=g); x = torch.randn(n, F, generator=g)
save('py_legacy_spmm.npz', index=index, value=value, m=m, n=n, mat=x, out=spmm(index, value, m, n, x))
for reduce in ('sum', 'mean', 'min', 'max'):
    key = torch.randperm(m * n, generator=g)[:nnz].sort().values
    row, col = key // n, key % n
    value = torch.randn(nnz, generator=g, dtype=torch.float64).requires_grad_()
    x = torch.randn(2, n, F, generator=g, dtype=torch.float64).requires_grad_()
    gout = torch.randn(2, m, F, generator=g, dtype=torch.float64)
    A = SparseTensor(row=row, col=col, value=value, sparse_sizes=(m, n))
    out = matmul(A, x, reduce)
    out.backward(gout)
    save('py_matmul_%s.npz' % reduce, row=row, col=col, value=value.detach(), mat=x.detach(), grad_out=gout, m=m, n=n,
         out=out.detach(), grad_value=value.grad, grad_mat=x.grad)
print('part 2: reference python fixtures written')
'''


PART3 = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
sys.path.insert(0, os.environ['TS_GOLDEN'])
import torch_sparse
import cases3
out_dir = os.environ['TS_OUT']
raw = cases3.make_inputs()
np.savez_compressed(os.path.join(out_dir, 'py3_inputs.npz'), **{k: v.numpy() for k, v in raw.items()})
I = cases3.tensors(torch_sparse, cases3.load_inputs(os.path.join(out_dir, 'py3_inputs.npz')), 'cpu')
n_ok, raised = 0, []
for name, fn in cases3.all_cases():
    try:
        out = fn(torch_sparse, I)
    except Exception as e:  # the reference itself rejects this call: no fixture
        raised.append('%s (%s)' % (name, type(e).__name__))
        continue
    d = {}
    if isinstance(out, torch.Tensor):
        d['dense'] = out.numpy()
    else:
        row, col, value = out.coo()
        d.update(row=row.numpy(), col=col.numpy(), sizes=np.array(out.sparse_sizes()),
                 rowptr=out.storage.rowptr().numpy())
        if value is not None:
            d['value'] = value.numpy()
    np.savez_compressed(os.path.join(out_dir, 'py3_%s.npz' % name), **d)
    n_ok += 1
print('part 3: %d fixtures; the reference raises for: %s' % (n_ok, ', '.join(raised) or '-'))
"""


def make_scratch(modules, op_sources):
    """A scratch copy of the reference Python package (symlinks + a loader __init__) next to an op
    library with the reference's own registration names, built straight from its sources."""
    from torch.utils import cpp_extension as ce
    scratch = tempfile.mkdtemp(prefix='ts_ref_py_')
    pkg = os.path.join(scratch, 'torch_sparse')
    os.makedirs(pkg)
    refpkg = os.path.join(REF, 'torch_sparse')
    for f in ['storage', 'tensor', 'utils', 'typing', 'testing'] + modules:
        os.symlink(os.path.join(refpkg, f + '.py'), os.path.join(pkg, f + '.py'))
    os.symlink(os.path.join(HERE, 'shims', 'torch_scatter'), os.path.join(scratch, 'torch_scatter'))
    csrc = os.path.join(REF, 'csrc')
    lib = os.path.join(pkg, '_ops_cpu.so')
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    inc = [csrc, os.path.join(ROOT, 'oracle', 'shim')] + ce.include_paths()
    srcs = [os.path.join(csrc, s) for s in op_sources]
    subprocess.check_call(['g++', '-O2', '-fopenmp', '-DAT_PARALLEL_OPENMP', '-Wno-sign-compare', '-std=c++17',
                           '-fPIC', '-shared', '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
                          + ['-I' + i for i in inc] + srcs +
                          ['-o', lib, '-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10', '-Wl,-rpath,' + tlib])
    return scratch, pkg


SPMM_SRCS = ('spmm.cpp', 'cpu/spmm_cpu.cpp', 'convert.cpp', 'cpu/convert_cpu.cpp')


def part2():
    scratch, pkg = make_scratch(['matmul', 'coalesce', 'transpose', 'spmm', 'spspmm'], SPMM_SRCS)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .transpose import t, transpose\nfrom .matmul import matmul\n"
                "from .coalesce import coalesce\nfrom .spmm import spmm\nfrom .spspmm import spspmm\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART2], env=env)


def part3():
    """py3_*.npz: narrow / select / index_select / masked_select / permute / cat / diag / add / mul of
    the reference Python package on the inputs and cases of tests/golden/cases3.py."""
    mods = ['transpose', 'coalesce', 'narrow', 'select', 'index_select', 'masked_select', 'permute', 'cat',
            'diag', 'add', 'mul', 'reduce']
    scratch, pkg = make_scratch(mods, SPMM_SRCS + ('diag.cpp', 'cpu/diag_cpu.cpp'))
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .transpose import t\nfrom .narrow import narrow, __narrow_diag__\n"
                "from .select import select\nfrom .index_select import index_select, index_select_nnz\n"
                "from .masked_select import masked_select, masked_select_nnz\nfrom .permute import permute\n"
                "from .diag import remove_diag, set_diag, fill_diag, get_diag\n"
                "from .add import add, add_, add_nnz, add_nnz_\nfrom .mul import mul, mul_, mul_nnz, mul_nnz_\n"
                "from .reduce import sum, mean, min, max\nfrom .cat import cat\n"
                "from .coalesce import coalesce\nfrom .transpose import transpose\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, TS_GOLDEN=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART3], env=env)


PART4 = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
import torch_sparse
from torch_sparse import SparseTensor
out_dir = os.environ['TS_OUT']
def save(name, **kw):
    np.savez_compressed(os.path.join(out_dir, name), **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})

g = torch.Generator().manual_seed(77)
n, nnz = 200, 1500
key = torch.cat([torch.randperm(n * n, generator=g)[:nnz], torch.arange(n) * n + (torch.arange(n) + 1) % n]).unique()
row, col = key // n, key % n                       # every node has at least one out-edge
value = torch.randint(-8, 9, (key.numel(),), generator=g).float() / 4
A = SparseTensor(row=row, col=col, value=value, sparse_sizes=(n, n))
rowptr = A.storage.rowptr()
save('py4_graph.npz', rowptr=rowptr, row=row, col=col, value=value, n=n)

# random walks: the op draws torch.rand((N, L)) from the global CPU generator -> re-draw it with the same seed
for seed, (N, L) in enumerate([(64, 10), (200, 3), (5, 0), (1, 40)]):
    start = torch.randint(0, n, (N,), generator=g)
    torch.manual_seed(100 + seed)
    out = torch.ops.torch_sparse.random_walk(rowptr, col, start, L)
    torch.manual_seed(100 + seed)
    rand = torch.rand((N, L))
    save('py4_rw_%d.npz' % seed, start=start, rand=rand, out=out)

# take-all neighbour "sampling" (num_neighbors = -1) is deterministic
for seed, m in enumerate([30, 1, 200, 0]):
    idx = torch.randperm(n, generator=g)[:m]
    r, c, n_id, e_id = torch.ops.torch_sparse.sample_adj(rowptr, col, idx, -1, False)
    adj, n_id2 = A.sample_adj(idx, -1)
    assert torch.equal(n_id, n_id2)
    save('py4_sample_all_%d.npz' % seed, idx=idx, rowptr=r, col=c, n_id=n_id, e_id=e_id, value=adj.storage.value())
    # oversized k without replacement also takes everything (perm = all positions), but the reference
    # walks a std::unordered_set there, so only the take-all case above has a defined n_id order

# SAINT sub-graphs and relabel
for seed, m in enumerate([50, 200, 1, 0]):
    idx = torch.randperm(n, generator=g)[:m]
    r, c, e = torch.ops.torch_sparse.saint_subgraph(idx, rowptr, row, col)
    sub, e2 = A.saint_subgraph(idx)
    assert torch.equal(e, e2)
    save('py4_saint_%d.npz' % seed, idx=idx, row=r, col=c, edge_index=e, value=sub.storage.value())
    sel = torch.randint(0, key.numel(), (3 * m,), generator=g)
    oc, oi = torch.ops.torch_sparse.relabel(col[sel], idx)
    save('py4_relabel_%d.npz' % seed, idx=idx, col=col[sel], out_col=oc, out_idx=oi)
    for bip in (False, True):
        orp, oc, ov, oi = torch.ops.torch_sparse.relabel_one_hop(rowptr, col, value, idx, bip)
        save('py4_relabel_one_hop_%d_%d.npz' % (seed, int(bip)), idx=idx, rowptr=orp, col=oc, value=ov, out_idx=oi)
print('part 4: sampler / walk / saint / relabel fixtures written')
"""


def part4():
    """py4_*.npz: random_walk (with the floats it drew), take-all sample_adj, saint_subgraph, relabel,
    relabel_one_hop of the reference's CPU kernels."""
    srcs = SPMM_SRCS + ('rw.cpp', 'cpu/rw_cpu.cpp', 'sample.cpp', 'cpu/sample_cpu.cpp', 'saint.cpp',
                        'cpu/saint_cpu.cpp', 'relabel.cpp', 'cpu/relabel_cpu.cpp')
    scratch, pkg = make_scratch(['transpose', 'coalesce', 'sample', 'saint', 'rw'], srcs)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .sample import sample, sample_adj\nfrom .saint import saint_subgraph\n"
                "from .rw import random_walk\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART4], env=env)


PART5 = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
import torch_sparse
out_dir = os.environ['TS_OUT']
G = np.load(os.path.join(out_dir, 'py4_graph.npz'))
n = int(G['n'])
# CSC view of the py4 graph: colptr over the columns, `row` holds the sources
order = np.lexsort((G['row'], G['col']))
row_csc, col_csc = G['row'][order], G['col'][order]
colptr = np.zeros(n + 1, np.int64); np.cumsum(np.bincount(col_csc, minlength=n), out=colptr[1:])
g = torch.Generator().manual_seed(5)
k = 0
for m in (1, 8, 40):
    inp = torch.randperm(n, generator=g)[:m]
    for fan in ([-1], [-1, -1], [-1, -1, -1], [1000, 1000]):
        for directed in (True, False):
            node, r, c, e = torch.ops.torch_sparse.neighbor_sample(torch.from_numpy(colptr), torch.from_numpy(row_csc), inp, fan, False, directed)
            np.savez_compressed(os.path.join(out_dir, 'py5_neighbor_sample_%02d.npz' % k), colptr=colptr, row=row_csc,
                                input_node=inp.numpy(), num_neighbors=np.array(fan), directed=directed,
                                node=node.numpy(), out_row=r.numpy(), out_col=c.numpy(), out_edge=e.numpy())
            k += 1
print('part 5: %d neighbor_sample fixtures written' % k)
"""


def part5():
    """py5_*.npz: the deterministic (take-all) cases of the reference's multi-hop neighbor_sample."""
    srcs = SPMM_SRCS + ('neighbor_sample.cpp', 'cpu/neighbor_sample_cpu.cpp')
    scratch, pkg = make_scratch([], srcs)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART5], env=env)


PART8 = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
import torch_sparse
out_dir = os.environ['TS_OUT']
g = torch.Generator().manual_seed(8)
node_types = ['paper', 'author', 'venue']
edge_types = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'in', 'venue'),
              ('venue', 'hosts', 'paper'), ('paper', 'by', 'author')]
k = 0
for sizes in ({'paper': 60, 'author': 25, 'venue': 4}, {'paper': 300, 'author': 120, 'venue': 9}):
    colptr, row = {}, {}
    for (s, r, d) in edge_types:
        rel = '__'.join((s, r, d))
        deg = torch.randint(0, 6, (sizes[d], ), generator=g)
        deg[::7] = 0
        cp = torch.zeros(sizes[d] + 1, dtype=torch.long); cp[1:] = deg.cumsum(0)
        colptr[rel] = cp
        row[rel] = torch.randint(0, sizes[s], (int(cp[-1]), ), generator=g)
    times = {t: torch.randint(0, 50, (sizes[t], ), generator=g) for t in node_types}
    for inputs in ({'paper': 5}, {'paper': 12, 'venue': 2}, {'author': 7}):
        inp = {t: torch.randperm(sizes[t], generator=g)[:m] for t, m in inputs.items()}
        for hops, fanval in ((1, -1), (2, -1), (3, -1), (2, 1000)):
            fan = {'__'.join(e): [fanval] * hops for e in edge_types}
            for mode in ('directed', 'undirected', 'temporal', 'temporal_partial'):
                if mode.startswith('temporal'):
                    tdict = dict(times) if mode == 'temporal' else {t: times[t] for t in ('paper', 'author') if True}
                    if mode == 'temporal_partial':  # a source type without time stamps is unconstrained
                        if 'venue' in inp:
                            continue
                        tdict = {t: times[t] for t in ('paper', 'author')}
                    out = torch.ops.torch_sparse.hetero_temporal_neighbor_sample(node_types, edge_types, colptr, row, inp, fan, tdict, hops, False, True)
                else:
                    tdict = {}
                    out = torch.ops.torch_sparse.hetero_neighbor_sample(node_types, edge_types, colptr, row, inp, fan, hops, False, mode == 'directed')
                blob = dict(mode=mode, hops=hops, fan=fanval)
                for rel in colptr:
                    blob['colptr__' + rel] = colptr[rel].numpy(); blob['row__' + rel] = row[rel].numpy()
                    blob['orow__' + rel] = out[1][rel].numpy(); blob['ocol__' + rel] = out[2][rel].numpy(); blob['oedge__' + rel] = out[3][rel].numpy()
                for t in node_types:
                    blob['node__' + t] = out[0][t].numpy()
                for t, x in inp.items():
                    blob['input__' + t] = x.numpy()
                for t, x in tdict.items():
                    blob['time__' + t] = x.numpy()
                np.savez_compressed(os.path.join(out_dir, 'py8_hetero_%03d.npz' % k), **blob)
                k += 1
print('part 8: %d hetero sampling fixtures written' % k)
"""


def part8():
    """py8_hetero_*.npz: the deterministic cases of the reference's heterogeneous (and temporal) neighbour samplers."""
    srcs = SPMM_SRCS + ('neighbor_sample.cpp', 'cpu/neighbor_sample_cpu.cpp')
    scratch, pkg = make_scratch([], srcs)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART8], env=env)


PART6 = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
sys.path.insert(0, os.environ['TS_GOLDEN'])
import torch_sparse
import cases6
out_dir = os.environ['TS_OUT']
N = int(os.environ.get('TS_NCASES', '672'))
blob, raised = {}, []
for i in range(N):
    c = cases6.make_case(i)
    try:
        outs = cases6.run_case(torch_sparse, c, 'cpu')
    except Exception as e:
        raised.append('%d:%s(%s)' % (i, c['op'], type(e).__name__))
        continue
    for k, v in c.items():
        blob['c%d_%s' % (i, k)] = np.asarray(v)
    for j, o in enumerate(outs):
        blob['o%d_%d' % (i, j)] = o
    blob['n%d' % i] = np.array(len(outs))
np.savez_compressed(os.path.join(out_dir, 'py6_random_cases.npz'), **blob)
print('part 6: %d random cases stored, the reference raised for %d: %s' % (N - len(raised), len(raised), ' '.join(raised)))
"""


def part6():
    """py6_random_cases.npz: several hundred randomised small cases (tests/golden/cases6.py) through the
    reference package."""
    mods = ['transpose', 'coalesce', 'narrow', 'select', 'index_select', 'masked_select', 'permute', 'cat',
            'diag', 'add', 'mul', 'reduce', 'matmul', 'sample', 'saint', 'spmm', 'spspmm']
    srcs = SPMM_SRCS + ('diag.cpp', 'cpu/diag_cpu.cpp', 'sample.cpp', 'cpu/sample_cpu.cpp', 'saint.cpp',
                        'cpu/saint_cpu.cpp')
    scratch, pkg = make_scratch(mods, srcs)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .transpose import t\nfrom .narrow import narrow, __narrow_diag__\n"
                "from .select import select\nfrom .index_select import index_select, index_select_nnz\n"
                "from .masked_select import masked_select, masked_select_nnz\nfrom .permute import permute\n"
                "from .diag import remove_diag, set_diag, fill_diag, get_diag\n"
                "from .add import add, add_, add_nnz, add_nnz_\nfrom .mul import mul, mul_, mul_nnz, mul_nnz_\n"
                "from .reduce import sum, mean, min, max\nfrom .matmul import matmul\nfrom .cat import cat\n"
                "from .sample import sample, sample_adj\nfrom .saint import saint_subgraph\n"
                "from .coalesce import coalesce\nfrom .transpose import transpose\n"
                "from .spmm import spmm\nfrom .spspmm import spspmm\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, TS_GOLDEN=HERE, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART6], env=env)


PART7 = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['TS_SCRATCH'])
from torch_sparse import spmm
import importlib.util                     # the input generator only, loaded by path: importing the package
spec = importlib.util.spec_from_file_location('synth', os.path.join(os.environ['TS_ROOT'], 'pytorch_sparse_amd', 'synth.py'))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)   # would register torch_sparse:: twice
m = n = 1000
row, col = synth.uniform_edges(m, n, 5000, seed=0)
index = torch.stack([row, col])
value = synth.values(5000, seed=1)
x = synth.features(n, 16, seed=2)
out = spmm(index, value, m, n, x)                        # torch_sparse/spmm.py:5-31
out64 = spmm(index, value.double(), m, n, x.double())
l1 = spmm(index, value.abs().double(), m, n, x.abs().double())
np.savez_compressed(os.path.join(os.environ['TS_OUT'], 'py7_c1_spmm.npz'), index=index.numpy(), value=value.numpy(),
                    mat=x.numpy(), m=m, n=n, out=out.numpy(), out64=out64.numpy(), l1=l1.numpy())
print('part 7: C1 fixture written; |out|max = %g' % float(out.abs().max()))
'''


def part7():
    scratch, pkg = make_scratch(['spmm'], SPMM_SRCS)
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write("import os, torch\n"
                "torch.ops.load_library(os.path.join(os.path.dirname(__file__), '_ops_cpu.so'))\n"
                "from .storage import SparseStorage\nfrom .tensor import SparseTensor\n"
                "from .spmm import spmm\n")
    env = dict(os.environ, TS_SCRATCH=scratch, TS_OUT=HERE, TS_ROOT=ROOT, OMP_NUM_THREADS='1')
    subprocess.check_call([sys.executable, '-c', PART7], env=env)


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference tree %s not present' % REF)
    todo = sys.argv[1:] or ['part1', 'part2', 'part3', 'part4', 'part5', 'part6', 'part7', 'part8']
    for name in todo:  # e.g. `make_golden.py part3` regenerates only the py3_* fixtures
        {'part1': part1, 'part2': part2, 'part3': part3, 'part4': part4, 'part5': part5, 'part6': part6,
         'part7': part7, 'part8': part8}[name]()

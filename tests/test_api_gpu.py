"""The reference's own test vectors and the reference-generated fixtures, run through the Python
API (SparseTensor / matmul / coalesce / transpose / spmm / spspmm) on the GPU, plus larger
randomized parity runs against the numpy oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as oc
from oracle import np_oracle as no

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def ts(dev):
    import pytorch_sparse_amd
    return pytorch_sparse_amd


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dtype is None else t.to(dev, dtype)


# ---- literals from the reference's tests ----------------------------------------------------------
def test_reference_storage_vectors(ts, dev):
    # test/test_storage.py:10-24, 46-92
    assert torch.ops.torch_sparse.ind2ptr(torch.tensor([2, 2, 4, 5, 5, 6], device=dev), 8).tolist() == \
        [0, 0, 0, 2, 2, 3, 5, 6, 6]
    row = torch.tensor([0, 0, 1, 1], device=dev)
    col = torch.tensor([0, 1, 0, 1], device=dev)
    st = ts.SparseStorage(row=row, col=col)
    assert st.rowcount().tolist() == [2, 2] and st.rowptr().tolist() == [0, 2, 4]
    assert st.colcount().tolist() == [2, 2] and st.colptr().tolist() == [0, 2, 4]
    assert st.csr2csc().tolist() == [0, 2, 1, 3] and st.csc2csr().tolist() == [0, 2, 1, 3]
    assert st.num_cached_keys() == 5
    # test/test_storage.py:27-43: sort on construct
    row = torch.tensor([0, 1, 0, 1], device=dev)
    col = torch.tensor([1, 0, 0, 1], device=dev)
    val = torch.tensor([1., 2., 3., 4.], device=dev)
    st = ts.SparseStorage(row=row, col=col, value=val)
    assert st.row().tolist() == [0, 0, 1, 1] and st.col().tolist() == [0, 1, 0, 1]
    assert st.value().tolist() == [3, 1, 2, 4]
    # test/test_storage.py:125-141: coalesce
    row = torch.tensor([0, 0, 0, 1, 1], device=dev)
    col = torch.tensor([0, 1, 1, 0, 1], device=dev)
    st = ts.SparseStorage(row=row, col=col, value=torch.tensor([1., 1, 1, 3, 4], device=dev))
    assert not st.is_coalesced()
    st = st.coalesce()
    assert st.is_coalesced()
    assert st.row().tolist() == [0, 0, 1, 1] and st.col().tolist() == [0, 1, 0, 1]
    assert st.value().tolist() == [1, 2, 3, 4]


def test_reference_coalesce_transpose_vectors(ts, dev):
    # test/test_coalesce.py:5-33 == README.md:132-152
    index = torch.tensor([[1, 0, 1, 0, 2, 1], [0, 1, 1, 1, 0, 0]], device=dev)
    value = torch.tensor([[1, 2], [2, 3], [3, 4], [4, 5], [5, 6], [6, 7]], device=dev)
    i, v = ts.coalesce(index, value, m=3, n=2)
    assert i.tolist() == [[0, 1, 1, 2], [1, 0, 1, 0]]
    assert v.tolist() == [[6, 8], [7, 9], [3, 4], [5, 6]]
    i, v = ts.coalesce(index, value, m=3, n=2, op='max')
    assert i.tolist() == [[0, 1, 1, 2], [1, 0, 1, 0]]
    assert v.tolist() == [[4, 5], [6, 7], [3, 4], [5, 6]]
    i, v = ts.coalesce(index, None, m=3, n=2)
    assert i.tolist() == [[0, 1, 1, 2], [1, 0, 1, 0]] and v is None
    # test/test_transpose.py:10-32 == README.md:177-197
    for dtype in (torch.half, torch.bfloat16, torch.float, torch.double, torch.int, torch.long):
        index = torch.tensor([[1, 0, 1, 0, 2, 1], [0, 1, 1, 1, 0, 0]], device=dev)
        value = torch.tensor([[1, 2], [2, 3], [3, 4], [4, 5], [5, 6], [6, 7]], dtype=dtype, device=dev)
        i, v = ts.transpose(index, value, m=3, n=2)
        assert i.tolist() == [[0, 0, 1, 1], [1, 2, 0, 1]]
        assert v.tolist() == [[7, 9], [5, 6], [6, 8], [3, 4]]
        index = torch.tensor([[1, 0, 1, 0], [0, 1, 1, 2]], device=dev)  # no duplicates
        i, v = ts.transpose(index, torch.tensor([1, 2, 3, 4], dtype=dtype, device=dev), m=2, n=3)
        assert i.tolist() == [[0, 1, 1, 2], [1, 0, 1, 0]] and v.tolist() == [1, 2, 3, 4]


def test_reference_spmm_spspmm_vectors(ts, dev):
    # test/test_spmm.py:10-19 == README.md:225-238
    index = torch.tensor([[0, 0, 1, 2, 2], [0, 2, 1, 0, 1]], device=dev)
    for dtype in (torch.half, torch.bfloat16, torch.float, torch.double, torch.int, torch.long):
        value = torch.tensor([1, 2, 4, 1, 3], dtype=dtype, device=dev)
        x = torch.tensor([[1, 4], [2, 5], [3, 6]], dtype=dtype, device=dev)
        assert ts.spmm(index, value, 3, 3, x).tolist() == [[7, 16], [8, 20], [7, 19]]
    # test/test_spspmm.py:10-22 == README.md:271-286
    for dtype in (torch.float, torch.double):
        iA = torch.tensor([[0, 0, 1, 2, 2], [1, 2, 0, 0, 1]], device=dev)
        vA = torch.tensor([1, 2, 3, 4, 5], dtype=dtype, device=dev)
        iB = torch.tensor([[0, 2], [1, 0]], device=dev)
        vB = torch.tensor([2, 4], dtype=dtype, device=dev)
        iC, vC = ts.spspmm(iA, vA, iB, vB, 3, 3, 2)
        assert iC.tolist() == [[0, 1, 2], [0, 1, 1]] and vC.tolist() == [8, 6, 8]
    # test/test_matmul.py:54-79: I @ I with / without values
    src = ts.SparseTensor.from_dense(torch.eye(3, device=dev))
    out = ts.matmul(src, src)
    assert out.sizes() == [3, 3] and out.has_value()
    rowptr, col, value = out.csr()
    assert rowptr.tolist() == [0, 1, 2, 3] and col.tolist() == [0, 1, 2] and value.tolist() == [1, 1, 1]
    src.set_value_(None)
    out = ts.matmul(src, src)
    assert not out.has_value() and out.csr()[1].tolist() == [0, 1, 2]
    # test/test_spspmm.py:25-51: orthonormal rows, x @ x^T == I through SpMM and through t() + SpSpMM
    for dtype in (torch.float, torch.double):
        x = ts.SparseTensor(
            row=torch.tensor([0, 1, 1, 1, 2, 3, 4, 5, 5, 6, 6, 7, 7, 7, 8, 8, 9, 9], device=dev),
            col=torch.tensor([0, 5, 10, 15, 1, 2, 3, 7, 13, 6, 9, 5, 10, 15, 11, 14, 5, 15], device=dev),
            value=torch.tensor([1, 3**-0.5, 3**-0.5, 3**-0.5, 1, 1, 1, -2**-0.5, -2**-0.5, -2**-0.5,
                                -2**-0.5, 6**-0.5, -6**0.5 / 3, 6**-0.5, -2**-0.5, -2**-0.5, 2**-0.5,
                                -2**-0.5], dtype=dtype, device=dev))
        expected = torch.eye(10, device=dev, dtype=dtype)
        assert torch.allclose(x @ x.to_dense().t(), expected, atol=1e-2)
        assert torch.allclose((x @ x.t()).to_dense(), expected, atol=1e-2)


@pytest.mark.parametrize('dtype', [torch.half, torch.bfloat16, torch.float, torch.double])
@pytest.mark.parametrize('reduce', ['sum', 'add', 'mean', 'min', 'max'])
def test_reference_matmul_autograd(ts, dev, dtype, reduce):
    """test/test_matmul.py:12-51 with an independent expected value: dense gather + scatter_reduce."""
    torch.manual_seed(0)
    src = torch.randn(10, 8, dtype=dtype, device=dev)
    src[2:4, :] = 0
    src[:, 2:4] = 0
    src = ts.SparseTensor.from_dense(src).requires_grad_()
    row, col, value = src.coo()
    other = torch.randn(2, 8, 2, dtype=dtype, device=dev, requires_grad=True)
    src_col = other.index_select(-2, col) * value.unsqueeze(-1)
    r = {'sum': 'sum', 'add': 'sum', 'mean': 'mean', 'min': 'amin', 'max': 'amax'}[reduce]
    idx = row.view(1, -1, 1).expand_as(src_col)
    expected = torch.zeros(2, 10, 2, dtype=dtype, device=dev).scatter_reduce(1, idx, src_col, r, include_self=False)
    grad_out = torch.randn_like(expected)
    expected.backward(grad_out)
    eg_value, eg_other = value.grad, other.grad
    value.grad = None
    other.grad = None
    out = ts.matmul(src, other, reduce)
    out.backward(grad_out)
    atol = 1e-1 if dtype in (torch.half, torch.bfloat16) else (1e-5 if dtype == torch.float else 1e-7)
    assert torch.allclose(expected, out, atol=atol)
    assert torch.allclose(eg_value, value.grad, atol=atol)
    assert torch.allclose(eg_other, other.grad, atol=atol)


# ---- fixtures generated by the reference's own Python ---------------------------------------------
def test_reference_python_fixtures(ts, dev):
    n = 0
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_coalesce_*.npz'))):
        z = np.load(f)
        value = T(z['value'], dev) if 'value' in z.files else None
        op = str(z['op']) if 'op' in z.files else 'add'
        i, v = ts.coalesce(T(z['index'], dev), value, int(z['m']), int(z['n']), op=op)
        assert np.array_equal(i.cpu().numpy(), z['out_index']), f
        if value is not None:
            assert np.array_equal(v.cpu().numpy(), z['out_value']), f
        n += 1
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_transpose_*.npz'))):
        z = np.load(f)
        i, v = ts.transpose(T(z['index'], dev), T(z['value'], dev), int(z['m']), int(z['n']))
        assert np.array_equal(i.cpu().numpy(), z['out_index']) and np.array_equal(v.cpu().numpy(), z['out_value']), f
        n += 1
    z = np.load(os.path.join(GOLDEN, 'py_storage.npz'))
    A = ts.SparseTensor(row=T(z['row'], dev), col=T(z['col'], dev), value=T(z['value'], dev),
                        sparse_sizes=(int(z['m']), int(z['n'])))
    r, c, v = A.coo()
    assert np.array_equal(r.cpu().numpy(), z['s_row']) and np.array_equal(c.cpu().numpy(), z['s_col'])
    assert np.array_equal(v.cpu().numpy(), z['s_value'])
    st = A.storage
    for key, fn in (('rowptr', st.rowptr), ('colptr', st.colptr), ('csr2csc', st.csr2csc),
                    ('csc2csr', st.csc2csr), ('rowcount', st.rowcount), ('colcount', st.colcount)):
        assert np.array_equal(fn().cpu().numpy(), z[key]), key
    tr, tc, tv = A.t().coo()
    assert np.array_equal(tr.cpu().numpy(), z['t_row']) and np.array_equal(tc.cpu().numpy(), z['t_col'])
    assert np.array_equal(tv.cpu().numpy(), z['t_value'])
    assert A.t().t() == A
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_spspmm_*.npz'))):
        z = np.load(f)
        m, k, nn = int(z['m']), int(z['k']), int(z['n'])
        if 'vA' in z.files:
            iC, vC = ts.spspmm(T(z['iA'], dev), T(z['vA'], dev), T(z['iB'], dev), T(z['vB'], dev), m, k, nn)
            assert np.array_equal(iC.cpu().numpy(), z['iC']), f
            assert np.allclose(vC.cpu().numpy(), z['vC'], rtol=1e-6, atol=1e-6), f
        else:
            A = ts.SparseTensor(row=T(z['iA'][0], dev), col=T(z['iA'][1], dev), sparse_sizes=(m, k))
            B = ts.SparseTensor(row=T(z['iB'][0], dev), col=T(z['iB'][1], dev), sparse_sizes=(k, nn))
            C = A @ B
            assert not C.has_value()
            rr, cc, _ = C.coo()
            assert np.array_equal(torch.stack([rr, cc]).cpu().numpy(), z['iC']), f
        n += 1
    z = np.load(os.path.join(GOLDEN, 'py_legacy_spmm.npz'))
    out = ts.spmm(T(z['index'], dev), T(z['value'], dev), int(z['m']), int(z['n']), T(z['mat'], dev))
    assert np.allclose(out.cpu().numpy(), z['out'], rtol=1e-5, atol=1e-5)
    for reduce in ('sum', 'mean', 'min', 'max'):
        z = np.load(os.path.join(GOLDEN, 'py_matmul_%s.npz' % reduce))
        value = T(z['value'], dev).requires_grad_()
        x = T(z['mat'], dev).requires_grad_()
        A = ts.SparseTensor(row=T(z['row'], dev), col=T(z['col'], dev), value=value,
                            sparse_sizes=(int(z['m']), int(z['n'])))
        out = ts.matmul(A, x, reduce)
        out.backward(T(z['grad_out'], dev))
        assert np.allclose(out.detach().cpu().numpy(), z['out'], rtol=1e-12, atol=1e-12), reduce
        assert np.allclose(value.grad.cpu().numpy(), z['grad_value'], rtol=1e-10, atol=1e-12), reduce
        assert np.allclose(x.grad.cpu().numpy(), z['grad_mat'], rtol=1e-10, atol=1e-12), reduce
        n += 1
    assert n > 30


# ---- randomized parity at larger sizes ------------------------------------------------------------
@pytest.mark.parametrize('n,m,ncols', [(0, 5, 5), (1, 1, 1), (1000, 7, 3), (300000, 5000, 4000),
                                       (3000000, 1 << 20, 1 << 20), (50000, 1 << 31, 1 << 30)])
def test_sort_coo_bit_exact(dev, n, m, ncols):
    g = torch.Generator().manual_seed(n % 97)
    row = torch.randint(0, m, (n, ), generator=g)
    col = torch.randint(0, ncols, (n, ), generator=g)
    if n >= 1000:  # force duplicates so that stability is exercised
        row[: n // 4] = row[n // 4: 2 * (n // 4)]
        col[: n // 4] = col[n // 4: 2 * (n // 4)]
    rs, cs, perm = torch.ops.tsamd.sort_coo(row.to(dev), col.to(dev), m, ncols, True)
    er, ec, ep = no.sort_coo(row.numpy(), col.numpy(), m, ncols)
    assert np.array_equal(perm.cpu().numpy(), ep)  # stable => the permutation itself is pinned
    assert np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)
    counts = torch.ops.tsamd.coo_order(rs, cs, ncols).tolist()
    assert counts[0] == 0
    key = er.astype(np.int64) * ncols + ec
    assert counts[1] == int((key[1:] == key[:-1]).sum())


@pytest.mark.parametrize('op', ['add', 'mean', 'min', 'max'])
def test_coalesce_large(ts, dev, op):
    g = torch.Generator().manual_seed(3)
    m, ncols, n = 20000, 300, 400000
    index = torch.stack([torch.randint(0, m, (n, ), generator=g), torch.randint(0, ncols, (n, ), generator=g)])
    for value in (torch.randint(-50, 50, (n, 3), generator=g), torch.randint(-64, 64, (n, ), generator=g).float() / 8,
                  torch.randint(-64, 64, (n, ), generator=g).double() / 8):
        i, v = ts.coalesce(index.to(dev), value.to(dev), m, ncols, op=op)
        er, ec, ev = no.coalesce(index[0].numpy(), index[1].numpy(), value.numpy(), m, ncols, op)
        assert np.array_equal(i.cpu().numpy(), np.stack([er, ec]))
        assert np.array_equal(v.cpu().numpy(), ev)  # dyadic values: every op is exact


def test_spspmm_vs_torch_sparse_mm(ts, dev):
    """Oracle = torch.sparse.mm on CPU, the very function behind the reference's spspmm."""
    g = torch.Generator().manual_seed(5)
    for (m, k, n, nA, nB, hub) in ((2000, 1500, 1800, 30000, 25000, False), (3000, 3000, 3000, 40000, 40000, True)):
        kA = torch.randperm(m * k, generator=g)[:nA].sort().values
        kB = torch.randperm(k * n, generator=g)[:nB].sort().values
        rA, cA, rB, cB = kA // k, kA % k, kB // n, kB % n
        if hub:  # a dense row of A and a dense row of B => rows beyond the LDS capacity
            rA = torch.cat([rA, torch.full((k, ), 7)]); cA = torch.cat([cA, torch.arange(k)])
            rB = torch.cat([rB, torch.full((n, ), 11)]); cB = torch.cat([cB, torch.arange(n)])
        for dtype in (torch.float32, torch.float64):
            vA = torch.randint(-4, 5, (rA.numel(), ), generator=g).to(dtype) / 2
            vB = torch.randint(-4, 5, (rB.numel(), ), generator=g).to(dtype) / 2
            A = torch.sparse_coo_tensor(torch.stack([rA, cA]), vA, (m, k)).coalesce()
            B = torch.sparse_coo_tensor(torch.stack([rB, cB]), vB, (k, n)).coalesce()
            C = torch.sparse.mm(A, B)
            iC, vC = ts.spspmm(A._indices().to(dev), A._values().to(dev), B._indices().to(dev),
                               B._values().to(dev), m, k, n)
            assert torch.equal(iC.cpu(), C._indices())
            assert torch.equal(vC.cpu(), C._values())  # half-integers: sums are exact in any order


def test_spspmm_small_rows_many_entries_and_empty_b_rows(ts, dev):
    """The one-wave expansion (expand_row_wave): rows of A with 1..200 entries -- several 64-entry chunks, products
    that start past a chunk's first batch -- against rows of B of 0 / 1 / 2 / 3 / 8 / 40 entries (zero-length rows
    between the others, batches of 256 products that begin inside a B row).  Oracle = torch.sparse.mm on the CPU;
    half-integer values keep every sum exact."""
    g = torch.Generator().manual_seed(11)
    m, k, n = 600, 900, 700
    lenB = torch.tensor([0, 0, 1, 2, 3, 8, 40])[torch.randint(0, 7, (k, ), generator=g)]
    rB = torch.repeat_interleave(torch.arange(k), lenB)
    cB = torch.cat([torch.randperm(n, generator=g)[:int(l)].sort().values for l in lenB.tolist()])
    lenA = torch.randint(1, 201, (m, ), generator=g)
    lenA[::7] = 0
    rA = torch.repeat_interleave(torch.arange(m), lenA)
    cA = torch.cat([torch.randperm(k, generator=g)[:int(l)].sort().values for l in lenA.tolist()])
    for dtype in (torch.float32, torch.float64):
        vA = torch.randint(-4, 5, (rA.numel(), ), generator=g).to(dtype) / 2
        vB = torch.randint(-4, 5, (rB.numel(), ), generator=g).to(dtype) / 2
        A = torch.sparse_coo_tensor(torch.stack([rA, cA]), vA, (m, k)).coalesce()
        B = torch.sparse_coo_tensor(torch.stack([rB, cB]), vB, (k, n)).coalesce()
        C = torch.sparse.mm(A, B)
        iC, vC = ts.spspmm(A._indices().to(dev), A._values().to(dev), B._indices().to(dev),
                           B._values().to(dev), m, k, n)
        assert torch.equal(iC.cpu(), C._indices())
        assert torch.equal(vC.cpu(), C._values())
    # pattern only (the value-less numeric kernel)
    At = ts.SparseTensor(row=rA.to(dev), col=cA.to(dev), sparse_sizes=(m, k))
    Bt = ts.SparseTensor(row=rB.to(dev), col=cB.to(dev), sparse_sizes=(k, n))
    Cp = At @ Bt
    row, col, val = Cp.coo()
    assert val is None
    assert torch.equal(torch.stack([row, col]).cpu(), C._indices())


def test_csr2csc_and_t_large(ts, dev):
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(16, 12, seed=9, device=dev)
    A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(c.numel(), device=dev),
                        sparse_sizes=(1 << 16, 1 << 16), is_sorted=True, trust_data=True)
    row, col, val = A.coo()
    p = no.csr2csc(row.cpu().numpy(), col.cpu().numpy(), 1 << 16, 1 << 16)
    assert np.array_equal(A.storage.csr2csc().cpu().numpy(), p)
    At = A.t()
    assert np.array_equal(At.storage.row().cpu().numpy(), col.cpu().numpy()[p])
    assert torch.equal(At.storage.rowptr(), A.storage.colptr())
    x = synth.features(1 << 16, 32, device=dev)
    dense_check = (At @ x)  # A^T x through the forward kernel on the transposed storage
    ref = torch.zeros_like(x).index_add_(0, col, val[:, None] * x[row])
    assert torch.allclose(dense_check, ref, rtol=1e-4, atol=1e-4)


def test_reductions_and_broadcast_mul(ts, dev):
    """reduce.py / mul.py against dense torch results; GCN normalisation."""
    g = torch.Generator().manual_seed(2)
    m, n = 700, 900
    dense = torch.randn(m, n, generator=g) * (torch.rand(m, n, generator=g) < 0.02)
    dense[100:140] = 0   # empty rows
    dense[:, 5:60] = 0   # empty columns
    A = ts.SparseTensor.from_dense(dense.to(dev))
    mask = dense != 0
    big = 1e30
    assert torch.allclose(A.sum(dim=1).cpu(), dense.sum(1), atol=1e-5)
    assert torch.allclose(A.sum(dim=0).cpu(), dense.sum(0), atol=1e-5)
    cnt1, cnt0 = mask.sum(1).clamp(min=1), mask.sum(0).clamp(min=1)
    assert torch.allclose(A.mean(dim=1).cpu(), dense.sum(1) / cnt1, atol=1e-5)
    assert torch.allclose(A.mean(dim=0).cpu(), dense.sum(0) / cnt0, atol=1e-5)
    mx1 = torch.where(mask, dense, torch.full_like(dense, -big)).max(1)[0]
    mn0 = torch.where(mask, dense, torch.full_like(dense, big)).min(0)[0]
    assert torch.equal(A.max(dim=1).cpu(), torch.where(mask.any(1), mx1, torch.zeros(m)))
    assert torch.equal(A.min(dim=0).cpu(), torch.where(mask.any(0), mn0, torch.zeros(n)))
    assert torch.allclose(A.sum().cpu(), dense.sum(), atol=1e-3)
    P = A.set_value(None)
    assert torch.equal(P.sum(dim=1).cpu(), mask.sum(1).float()) and torch.equal(P.sum(dim=0).cpu(), mask.sum(0).float())
    # D^-1/2 A D^-1/2 through mul(row-wise) and mul(col-wise), then SpMM
    Pn = P.fill_value(1.)
    deg = Pn.sum(dim=1)
    dinv = deg.clamp(min=1).pow(-0.5)
    Asq = ts.SparseTensor.from_dense((mask[:, :m]).float().to(dev))
    d = Asq.sum(dim=1).clamp(min=1).pow(-0.5)
    norm = ts.mul(ts.mul(Asq, d.view(-1, 1)), d.view(1, -1))
    ref = d.cpu()[:, None] * mask[:, :m].float() * d.cpu()[None, :]
    assert torch.allclose(norm.to_dense().cpu(), ref, atol=1e-6)
    x = torch.randn(m, 16, generator=g)
    assert torch.allclose((norm @ x.to(dev)).cpu(), ref @ x, atol=1e-4)
    assert dinv.numel() == m


def test_spspmm_rmat_all_size_classes(ts, dev):
    """A * A^T on an R-MAT graph: hub rows go through the global expand/sort path, mid rows through
    the 256-thread LDS kernel, the rest through the one-wave radix-sort kernel.  Oracle =
    torch.sparse.mm on CPU; integer-valued entries keep every sum exact."""
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(13, 8, seed=4, device=dev)
    n = 1 << 13
    g = torch.Generator(device=dev).manual_seed(1)
    val = torch.randint(1, 4, (c.numel(), ), generator=g, device=dev).float()
    A = ts.SparseTensor(rowptr=rp, col=c, value=val, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    At = A.t()
    rpB = At.storage.rowptr()
    prod = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, A.storage.row(), (rpB[1:] - rpB[:-1])[c])
    assert int((prod <= 512).sum()) > 0 and int(((prod > 512) & (prod <= 4096)).sum()) > 0 and int((prod > 4096).sum()) > 0
    C = A @ At
    Cc = torch.sparse.mm(A.cpu().to_torch_sparse_coo_tensor(), At.cpu().to_torch_sparse_coo_tensor())
    row, col, v = C.coo()
    assert torch.equal(torch.stack([row, col]).cpu(), Cc._indices())
    assert torch.equal(v.cpu(), Cc._values())
    assert C.is_coalesced()


def test_spspmm_values_are_reproducible_run_to_run(ts, dev):
    """Rounding-sensitive fp32 values (not the exact half-integers of the tests above) on a product whose hub rows fill
    (row, range) bins of far more than 1024 products -- the dense accumulation of the large-row path.  Round 4 summed
    those bins in atomic arrival order (not reproducible run to run, unlike the reference's CPU
    path); since round 5 a bin's products lie in a fixed order (per-wave segments) and one wave adds them in that
    order: five runs, bit-identical values; and they agree with torch.sparse.mm within 1e-5 of the L1 mass."""
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(15, 12, seed=9, device=dev)
    n = 1 << 15
    val = synth.values(c.numel(), seed=3, device=dev) - 0.5
    A = ts.SparseTensor(rowptr=rp, col=c, value=val, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    At = A.t()
    rpB = At.storage.rowptr()
    prod = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, A.storage.row(), (rpB[1:] - rpB[:-1])[c])
    assert int(prod.max()) > 200000  # hub rows: bins of >> 1024 products (4 column ranges at this size)
    first = A @ At
    r0, c0, v0 = first.coo()
    for _ in range(4):
        again = A @ At
        r1, c1, v1 = again.coo()
        assert torch.equal(r0, r1) and torch.equal(c0, c1)
        assert torch.equal(v0.view(torch.int32), v1.view(torch.int32))
    Cc = torch.sparse.mm(A.cpu().to_torch_sparse_coo_tensor(), At.cpu().to_torch_sparse_coo_tensor()).coalesce()
    assert torch.equal(torch.stack([r0, c0]).cpu(), Cc._indices())
    l1 = (A.set_value(val.abs(), 'coo') @ At.set_value(At.storage.value().abs(), 'coo')).storage.value()
    assert bool(((v0.cpu().double() - Cc._values().double()).abs() <= 1e-5 * l1.cpu().double() + 1e-30).all())


def test_fuzz_coalesce_transpose_spspmm(ts, dev):
    """Random small COO inputs (empty, single entry, all duplicates, tall/wide) against the numpy
    oracle: indices bit-exact, values exact (small integers)."""
    rng = np.random.RandomState(7)
    for case in range(120):
        m, n = int(rng.choice([1, 2, 9, 100, 1000])), int(rng.choice([1, 3, 50, 2000]))
        nnz = int(rng.choice([0, 1, 2, 17, 300, 3000]))
        row = torch.from_numpy(rng.randint(0, m, size=nnz)).long()
        col = torch.from_numpy(rng.randint(0, n, size=nnz)).long()
        if rng.rand() < 0.2 and nnz > 0:
            row[:] = row[0]
            col[:] = col[0]
        val = torch.from_numpy(rng.randint(-9, 10, size=nnz)).float()
        op = ['add', 'mean', 'min', 'max'][rng.randint(4)]
        if op == 'mean':
            val = val * 0 + 2.0  # keep the mean exact
        index = torch.stack([row, col])
        i, v = ts.coalesce(index.to(dev), val.to(dev), m, n, op=op)
        er, ec, ev = no.coalesce(row.numpy(), col.numpy(), val.numpy(), m, n, op)
        assert np.array_equal(i.cpu().numpy(), np.stack([er, ec])), case
        assert np.array_equal(v.cpu().numpy(), ev), case
        ti, tv = ts.transpose(index.to(dev), val.to(dev), m, n)
        tr, tc, tvv = no.transpose(row.numpy(), col.numpy(), val.numpy(), m, n)
        assert np.array_equal(ti.cpu().numpy(), np.stack([tr, tc])) and np.array_equal(tv.cpu().numpy(), tvv), case
        if case % 3 == 0 and er.size > 0:
            # C = A * A^T with the coalesced pattern
            A = ts.SparseTensor(row=i[0], col=i[1], value=v, sparse_sizes=(m, n), is_sorted=True, trust_data=True)
            C = A @ A.t()
            cr, cc, cv = C.coo()
            xr, xc, xv = no.spspmm(er, ec, ev, ec[np.argsort(ec * m + er, kind='stable')],
                                   er[np.argsort(ec * m + er, kind='stable')],
                                   ev[np.argsort(ec * m + er, kind='stable')], m, n, m)
            assert np.array_equal(torch.stack([cr, cc]).cpu().numpy(), np.stack([xr, xc])), case
            assert np.array_equal(cv.cpu().numpy(), xv.astype(np.float32)), case


def test_cpu_built_tensor_is_sorted_on_the_gpu(ts, dev):
    row = torch.tensor([2, 0, 1, 0, 2]); col = torch.tensor([1, 2, 0, 0, 0]); val = torch.tensor([1., 2, 3, 4, 5])
    A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(3, 3))     # CPU, unsorted: deferred
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        A.csr()
    G = A.to(dev)
    r, c, v = G.coo()
    assert r.tolist() == [0, 0, 1, 2, 2] and c.tolist() == [0, 2, 0, 0, 1] and v.tolist() == [4, 2, 3, 5, 1]
    assert G.storage.rowptr().tolist() == [0, 2, 3, 5]
    assert torch.equal((G @ torch.eye(3, device=dev)).cpu(), torch.tensor([[4., 0, 2], [3, 0, 0], [5, 1, 0]]))
    back = G.cpu()                       # sorted data may live on the CPU again
    assert back.storage.col().tolist() == [0, 2, 0, 0, 1]


# ---------------------------------------------------------------------------------------------
# reductions and coalesce are differentiable w.r.t. the sparse values (torch_scatter's segment_csr
# is, in the reference): compare with torch's own scatter autograd
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
@pytest.mark.parametrize('dim', [0, 1])
def test_reduction_gradients(reduce, dim):
    import pytorch_sparse_amd as ts
    g = torch.Generator().manual_seed(3)
    m, n, nnz = 300, 200, 4000
    key = torch.randperm(m * n, generator=g)[:nnz].sort().values
    row, col = (key // n).cuda(), (key % n).cuda()
    for shape in ((nnz, ), (nnz, 3)):
        v0 = torch.randn(shape, generator=g, dtype=torch.float64).cuda()  # no ties
        value = v0.clone().requires_grad_()
        A = ts.SparseTensor(row=row, col=col, value=value, sparse_sizes=(m, n), is_sorted=True)
        out = getattr(A, reduce)(dim=dim)
        gout = torch.randn(out.shape, generator=g, dtype=torch.float64).cuda()
        out.backward(gout)
        # reference: torch.scatter_reduce along the kept index
        ref_v = v0.clone().requires_grad_()
        index = row if dim == 1 else col
        size = m if dim == 1 else n
        idx = index.view((-1, ) + (1, ) * (ref_v.dim() - 1)).expand_as(ref_v)
        red = {'sum': 'sum', 'mean': 'mean', 'min': 'amin', 'max': 'amax'}[reduce]
        want = torch.zeros((size, ) + tuple(shape[1:]), dtype=torch.float64, device='cuda').scatter_reduce(
            0, idx, ref_v, red, include_self=False)
        want.backward(gout)
        torch.testing.assert_close(out.detach(), want.detach(), rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(value.grad, ref_v.grad, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('op', ['add', 'mean', 'min', 'max'])
def test_coalesce_gradients(op):
    import pytorch_sparse_amd as ts
    g = torch.Generator().manual_seed(4)
    m, n, nnz = 40, 30, 3000  # many duplicates
    index = torch.stack([torch.randint(0, m, (nnz, ), generator=g), torch.randint(0, n, (nnz, ), generator=g)]).cuda()
    v0 = torch.randn(nnz, 2, generator=g, dtype=torch.float64).cuda()
    value = v0.clone().requires_grad_()
    oi, ov = ts.coalesce(index, value, m, n, op=op)
    gout = torch.randn(ov.shape, generator=g, dtype=torch.float64).cuda()
    ov.backward(gout)
    ref_v = v0.clone().requires_grad_()
    key = index[0] * n + index[1]
    uniq, inv = key.unique(return_inverse=True)
    red = {'add': 'sum', 'mean': 'mean', 'min': 'amin', 'max': 'amax'}[op]
    want = torch.zeros((uniq.numel(), 2), dtype=torch.float64, device='cuda').scatter_reduce(
        0, inv.view(-1, 1).expand(-1, 2), ref_v, red, include_self=False)
    want.backward(gout)
    assert torch.equal(oi[0] * n + oi[1], uniq)
    torch.testing.assert_close(ov.detach(), want.detach(), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(value.grad, ref_v.grad, rtol=1e-12, atol=1e-12)
    # SparseTensor.coalesce and sparse + sparse carry the gradient too
    A = ts.SparseTensor(row=index[0], col=index[1], value=value, sparse_sizes=(m, n))
    value.grad = None
    A.coalesce(reduce=op if op != 'add' else 'sum').storage.value().sum().backward()
    assert value.grad is not None and float(value.grad.abs().sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64, torch.bfloat16, torch.int64, torch.int32])
def test_reductions_hub_rows_balanced_path(ts, dev, dtype):
    """Row / column reductions of a power-law matrix run on the entry-balanced path (A * 1 on the SpMM
    kernel).  Small-integer values make every sum exact in every dtype, so the result must equal
    both torch.scatter_reduce and the thread-per-segment kernel bit for bit (means: floor for ints)."""
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(16, 20, seed=1)
    n, E = rp.numel() - 1, c.numel()
    assert E > 32768 and int((rp[1:] - rp[:-1]).max()) > 2000  # hubs present, balanced path taken
    g = torch.Generator().manual_seed(0)
    for D in (1, 3):
        shape = (E, ) if D == 1 else (E, D)
        v64 = torch.randint(-8, 9, shape, generator=g)
        value = v64.to(dtype).to(dev)
        A = ts.SparseTensor(rowptr=rp.to(dev), col=c.to(dev), value=value, sparse_sizes=(n, n), is_sorted=True)
        st = A.storage
        for dim in (1, 0):
            ptr = st.rowptr() if dim == 1 else st.colptr()
            perm = None if dim == 1 else st.csr2csc()
            index = (st.row() if dim == 1 else st.col())
            idx = index.view((-1, ) + (1, ) * (D > 1)).expand(shape)
            cnt = (ptr[1:] - ptr[:-1])
            for reduce in ('sum', 'mean', 'min', 'max'):
                got = getattr(A, reduce)(dim=dim)
                slow = torch.ops.tsamd.segment_reduce(value, perm, ptr, n, reduce, False)
                red = {'sum': 'sum', 'mean': 'sum', 'min': 'amin', 'max': 'amax'}[reduce]
                ref = torch.zeros((n, ) + shape[1:], dtype=torch.float64, device=dev).scatter_reduce(
                    0, idx, v64.to(dev).double(), red, include_self=False)
                if reduce == 'mean':
                    c_ = cnt.clamp(min=1).view((-1, ) + (1, ) * (D > 1)).double()
                    ref = torch.floor(ref / c_) if not dtype.is_floating_point else ref / c_
                if reduce == 'mean' and dtype.is_floating_point:
                    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
                    assert torch.allclose(got.double(), ref, rtol=tol, atol=tol), (dim, reduce, D)
                    assert torch.allclose(got.double(), slow.double(), rtol=tol, atol=tol)
                else:
                    if dtype == torch.bfloat16 and reduce == 'sum':  # hub sums exceed bf16's 8 bits
                        assert torch.allclose(got.double(), ref, rtol=1e-2, atol=1.0)
                        continue
                    assert torch.equal(got.double(), ref), (dim, reduce, D, dtype)
                    assert torch.equal(got, slow), (dim, reduce, D, dtype)
    # gradients flow through the balanced path too
    val = torch.randn(E, generator=g, dtype=torch.float64).to(dev).requires_grad_()
    A = ts.SparseTensor(rowptr=rp.to(dev), col=c.to(dev), value=val, sparse_sizes=(n, n), is_sorted=True)
    gout = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    for reduce, red in (('sum', 'sum'), ('max', 'amax')):
        val.grad = None
        getattr(A, reduce)(dim=1).backward(gout)
        ref_v = val.detach().clone().requires_grad_()
        torch.zeros(n, dtype=torch.float64, device=dev).scatter_reduce(0, A.storage.row(), ref_v, red,
                                                                         include_self=False).backward(gout)
        torch.testing.assert_close(val.grad, ref_v.grad, rtol=1e-12, atol=1e-12)


def test_coalesce_heavy_duplication(ts, dev):
    """A handful of distinct pairs repeated a million times: the duplicate runs are long, the values
    are reduced on the entry-balanced path; exact with small integers."""
    g = torch.Generator().manual_seed(9)
    nnz = 1_500_000
    row = torch.randint(0, 3, (nnz, ), generator=g)
    col = torch.randint(0, 2, (nnz, ), generator=g)
    val = torch.randint(-4, 5, (nnz, 2), generator=g).float()
    for op in ('add', 'mean', 'min', 'max'):
        oi, ov = ts.coalesce(torch.stack([row, col]).to(dev), val.to(dev), 3, 2, op=op)
        key = row * 2 + col
        uniq, inv = key.unique(return_inverse=True)
        red = {'add': 'sum', 'mean': 'mean', 'min': 'amin', 'max': 'amax'}[op]
        want = torch.zeros(uniq.numel(), 2, dtype=torch.float64).scatter_reduce(
            0, inv.view(-1, 1).expand(-1, 2), val.double(), red, include_self=False)
        assert torch.equal((oi[0] * 2 + oi[1]).cpu(), uniq)
        assert torch.allclose(ov.cpu().double(), want, rtol=1e-6, atol=1e-6), op
    A = ts.SparseTensor(row=row.to(dev), col=col.to(dev), value=val[:, 0].contiguous().to(dev), sparse_sizes=(3, 2))
    C = A.coalesce('sum')
    assert C.nnz() == uniq.numel()
    want = torch.zeros(uniq.numel(), dtype=torch.float64).scatter_reduce(0, inv, val[:, 0].double(), 'sum')
    assert torch.equal(C.storage.value().cpu().double(), want)


@pytest.mark.parametrize('P', [2, 4, 8])
def test_row_shards_as_logical_ranks_on_one_gpu(ts, dev, P):
    """SURVEY.md 8e validation: the sharded path with P logical ranks run one after the other on one
    device.  Every rank owns a row block of A (narrow_rows, nnz-balanced) with global column ids and
    (i) multiplies it with the full X, (ii) multiplies the column-compacted block with only the rows
    of X it references (what the halo exchange delivers).  Stacked, both must equal the unsharded
    SpMM: bit for bit for max, and to fp32 rounding for sum / mean (where a long row is cut by the
    merge-path partition depends on the block it sits in, so its partial sums associate differently)."""
    from pytorch_sparse_amd import _native as nat
    from pytorch_sparse_amd import synth
    from pytorch_sparse_amd.parallel import narrow_rows, partition_rows
    rp, c = synth.rmat_csr(15, 16, seed=4)
    n, E = rp.numel() - 1, c.numel()
    rp, c = rp.to(dev), c.to(dev)
    v = synth.values(E, device=dev)
    x = synth.features(n, 48, device=dev)
    ranges = partition_rows(rp, P, 'nnz')
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    nnz = [int(rp[e] - rp[s]) for s, e in ranges]
    assert max(nnz) - min(nnz) <= int((rp[1:] - rp[:-1]).max())  # balanced up to one row
    for reduce in ('sum', 'mean', 'max'):
        full = nat.spmm(rp, c, v, x, reduce)[0]
        direct, halo = [], []
        for s, e in ranges:
            rpl, cl, vl = narrow_rows(rp, c, v, s, e)
            direct.append(nat.spmm(rpl.contiguous(), cl, vl, x, reduce)[0])
            needed = torch.unique(cl)                      # rows of X this block references
            compact = torch.searchsorted(needed, cl)       # column ids = positions in the fetched rows
            halo.append(nat.spmm(rpl.contiguous(), compact, vl, x[needed].contiguous(), reduce)[0])
        if reduce == 'max':
            assert torch.equal(torch.cat(direct), full) and torch.equal(torch.cat(halo), full)
        else:
            assert torch.allclose(torch.cat(direct), full, rtol=1e-5, atol=1e-5), reduce
            assert torch.equal(torch.cat(halo), torch.cat(direct)), reduce  # same blocks, relabelled columns


# ---- construction without host syncs ---------------------------------------------
@pytest.mark.parametrize('presorted', [False, True])
def test_constructor_is_sync_free_with_trusted_data(dev, ts, presorted):
    """SparseTensor(row, col, value, sparse_sizes, trust_data=True) on device tensors reads nothing back: the sort
    is decided on the device (tsamd::sort_coo_auto).  torch's sync debug mode turns any synchronising call
    (.item(), .tolist(), .cpu(), nonzero ...) into an error; the result equals the synchronising constructor's."""
    from pytorch_sparse_amd import synth
    m, n, nnz = 3000, 2500, 200000
    row, col = synth.uniform_edges(m, n, nnz, seed=3, device=dev)
    val = synth.values(nnz, device=dev)
    if presorted:
        perm = torch.argsort(row * n + col, stable=True)
        row, col, val = row[perm], col[perm], val[perm]
    ref = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n))  # one sync (range check + order probe)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode('error')
    try:
        with pytest.raises(RuntimeError):  # the detector works on this build: a read-back raises
            int(row[0])
        A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n), trust_data=True)
        rowptr = A.storage.rowptr()
        out = A.matmul(synth.features(n, 8, device=dev))  # and the product right behind it, still no sync
    finally:
        torch.cuda.set_sync_debug_mode('default')
    r0, c0, v0 = ref.coo()
    r1, c1, v1 = A.coo()
    assert torch.equal(r0, r1) and torch.equal(c0, c1) and torch.equal(v0, v1)
    assert torch.equal(rowptr, ref.storage.rowptr())
    assert torch.equal(out, ref.matmul(synth.features(n, 8, device=dev)))


def test_constructor_single_sync_checks_and_infers(dev, ts):
    """Without trust_data the range check, the size inference and the order probe share one read-back
    (tsamd::coo_check); out-of-range ids still raise, sizes are still inferred."""
    row = torch.tensor([3, 0, 2, 0], device=dev)
    col = torch.tensor([1, 5, 0, 2], device=dev)
    A = ts.SparseTensor(row=row, col=col)
    assert A.sparse_sizes() == (4, 6)
    assert A.storage.row().tolist() == [0, 0, 2, 3] and A.storage.col().tolist() == [2, 5, 0, 1]
    with pytest.raises(Exception):
        ts.SparseTensor(row=row, col=col, sparse_sizes=(3, 6))
    with pytest.raises(Exception):
        ts.SparseTensor(row=row, col=col, sparse_sizes=(4, 5))
    counts = torch.ops.tsamd.coo_check(row, col).tolist()
    assert counts == [2, 0, 3, 5]

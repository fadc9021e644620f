"""Parity of the HIP hot path against the oracle, through the C-ABI (pytorch_sparse_amd._native).
All tests here need a real MI355X."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as oc
from pytorch_sparse_amd import _native as nat
from pytorch_sparse_amd import synth
from tests.util import (ALL_DTYPES, CODE, FLOAT_DTYPES, SUM_ATOL, SUM_TOL, bits_equal, check_spmm, experiments_build, fromnp,
                        oracle_spmm, tonp)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def make_inputs(rp, c, n, K, dtype, has_value, batch=(), seed=0):
    E = c.numel()
    if dtype.is_floating_point:
        v = synth.values(E, seed=seed + 1, dtype=dtype) - 0.3 if has_value else None
        x = synth.features(n, K, seed=seed + 2, dtype=dtype, batch=batch)
    else:
        g = torch.Generator().manual_seed(seed)
        lo = 0 if dtype == torch.uint8 else 1  # unsigned: no negative draws (sums still wrap)
        v = torch.randint(-4 * lo, 5, (E, ), dtype=dtype, generator=g) if has_value else None
        x = torch.randint(-9 * lo, 9 + 100 * (1 - lo), (*batch, n, K), dtype=dtype, generator=g)
    return v, x


def run_gpu(dev, rp, c, v, x, reduce):
    out, arg = nat.spmm(rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev), reduce)
    torch.cuda.synchronize()
    return out, arg


@pytest.mark.parametrize('dtype', ALL_DTYPES)
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
def test_spmm_rmat_all_dtypes(dev, dtype, reduce):
    rp, c = synth.rmat_csr(10, 16, seed=0)  # max degree 354, 128-item partitions => cut rows
    for K, has_value, batch in ((128, True, ()), (16, False, ()), (3, True, (2, )), (64, True, ())):
        v, x = make_inputs(rp, c, 1 << 10, K, dtype, has_value, batch)
        out, arg = run_gpu(dev, rp, c, v, x, reduce)
        check_spmm(out, arg, rp, c, v, x, reduce)


@pytest.mark.parametrize('K', [1, 2, 5, 32, 100, 256, 512, 1000])
@pytest.mark.parametrize('reduce', ['sum', 'max'])
def test_spmm_feature_widths(dev, K, reduce):
    rp, c = synth.rmat_csr(11, 12, seed=1)
    v, x = make_inputs(rp, c, 1 << 11, K, torch.float32, True)
    out, arg = run_gpu(dev, rp, c, v, x, reduce)
    check_spmm(out, arg, rp, c, v, x, reduce)


def test_reference_test_shape(dev):
    # test/test_matmul.py:18-25 of the reference: 10x8 with empty rows 2:4 / cols 2:4, other [2,8,2]
    torch.manual_seed(0)
    src = torch.randn(10, 8)
    src[2:4, :] = 0
    src[:, 2:4] = 0
    row, col = src.nonzero().t()
    val = src[row, col]
    rp = torch.from_numpy(oc.ind2ptr(row.numpy(), 10))
    for dtype in FLOAT_DTYPES:
        other = torch.randn(2, 8, 2).to(dtype)
        for reduce in ('sum', 'mean', 'min', 'max'):
            out, arg = run_gpu(dev, rp, col, val.to(dtype), other, reduce)
            assert out.shape == (2, 10, 2)
            check_spmm(out, arg, rp, col, val.to(dtype), other, reduce)


def test_empty_and_degenerate(dev):
    # E == 0: every row empty -> zeros, arg == E == 0
    rp = torch.zeros(6, dtype=torch.int64)
    c = torch.zeros(0, dtype=torch.int64)
    x = torch.randn(4, 8)
    for reduce in ('sum', 'mean', 'min', 'max'):
        out, arg = run_gpu(dev, rp, c, None, x, reduce)
        assert out.shape == (5, 8) and (out == 0).all()
        if arg is not None:
            assert (arg == 0).all()
    # M == 0
    out, _ = run_gpu(dev, torch.zeros(1, dtype=torch.int64), c, None, x, 'sum')
    assert out.shape == (0, 8)
    # K == 0
    out, _ = run_gpu(dev, rp, c, None, torch.randn(4, 0), 'sum')
    assert out.shape == (5, 0)


def test_single_hub_row_and_many_empty(dev):
    # one row with 100k entries (cut into ~100+ partitions) between empty rows, plus a long tail
    n = 5000
    g = torch.Generator().manual_seed(3)
    hub = torch.randint(0, n, (100000, ), generator=g)
    deg = torch.zeros(3000, dtype=torch.int64)
    deg[1234] = hub.numel()
    deg[2000:2100] = 7
    rp = torch.zeros(3001, dtype=torch.int64)
    torch.cumsum(deg, 0, out=rp[1:])
    c = torch.cat([hub, torch.randint(0, n, (700, ), generator=g)])
    for dtype in (torch.float32, torch.bfloat16):
        v, x = make_inputs(rp, c, n, 64, dtype, True)
        for reduce in ('sum', 'mean', 'min', 'max'):
            out, arg = run_gpu(dev, rp, c, v, x, reduce)
            check_spmm(out, arg, rp, c, v, x, reduce)


def test_ties_nan_and_no_winner(dev):
    rp = torch.tensor([0, 3, 3, 5])
    c = torch.tensor([0, 1, 2, 0, 0])
    x = torch.tensor([[1., 5.], [1., 7.], [1., 7.]])
    out, arg = run_gpu(dev, rp, c, None, x, 'max')
    assert out.tolist() == [[1, 7], [0, 0], [1, 5]] and arg.tolist() == [[0, 1], [5, 5], [3, 3]]
    out, arg = run_gpu(dev, rp, c, None, x, 'min')
    assert arg.tolist() == [[0, 0], [5, 5], [3, 3]]
    # many equal values across partitions / groups: the smallest edge id must win
    E = 3000
    rp = torch.tensor([0, E])
    c = torch.zeros(E, dtype=torch.int64)
    x = torch.full((1, 16), 2.5)
    for reduce in ('min', 'max'):
        out, arg = run_gpu(dev, rp, c, None, x, reduce)
        assert (arg == 0).all() and (out == 2.5).all()
    # NaN never wins, propagates through sum
    x = torch.tensor([[1.0], [float('nan')], [3.0]])
    out, arg = run_gpu(dev, torch.tensor([0, 3]), torch.tensor([0, 1, 2]), None, x, 'max')
    assert out.item() == 3.0 and arg.item() == 2
    out, _ = run_gpu(dev, torch.tensor([0, 3]), torch.tensor([0, 1, 2]), None, x, 'sum')
    assert torch.isnan(out).all()
    # a row of NaNs only: value stays at the reducer's init, arg reports E (documented divergence:
    # the reference leaves a stale index there)
    x = torch.full((2, 4), float('nan'))
    out, arg = run_gpu(dev, torch.tensor([0, 2]), torch.tensor([0, 1]), None, x, 'max')
    assert (arg == 2).all() and (out == torch.finfo(torch.float32).min).all()


def test_golden_fixtures_on_gpu(dev):
    files = sorted(glob.glob(os.path.join(GOLDEN, 'spmm_*.npz')))
    assert files
    inv = {v: k for k, v in CODE.items()}
    for f in files:
        z = np.load(f)
        dtype, reduce = inv[int(z['dtype_code'])], str(z['reduce'])
        rp, c = torch.from_numpy(z['rowptr']), torch.from_numpy(z['col'])
        x = fromnp(z['mat'], dtype)
        v = fromnp(z['value'], dtype) if 'value' in z.files else None
        out, arg = run_gpu(dev, rp, c, v, x, reduce)
        exact = reduce in ('min', 'max') or not dtype.is_floating_point
        if exact:
            assert bits_equal(out, fromnp(z['out'], dtype)), f
            if arg is not None:
                assert torch.equal(arg.cpu(), torch.from_numpy(z['arg_out'])), f
        elif dtype in (torch.float32, torch.float64):
            ref = fromnp(z['out'], dtype).double()
            tol = 1e-5 if dtype == torch.float32 else 1e-12
            assert torch.allclose(out.cpu().double(), ref, rtol=tol, atol=tol), f
        else:  # narrow sum/mean: fp32 accumulation by design, checked against the wide oracle
            check_spmm(out, arg, rp, c, v, x, reduce)


@pytest.mark.parametrize('dtype', FLOAT_DTYPES)
@pytest.mark.parametrize('reduce', ['sum', 'mean'])
def test_value_bw(dev, dtype, reduce):
    rp, c = synth.rmat_csr(10, 12, seed=5)
    n, E = 1 << 10, c.numel()
    row = torch.from_numpy(oc.ptr2ind(rp.numpy(), E))
    for K, batch in ((128, ()), (20, (2, )), (3, ())):
        x = synth.features(n, K, seed=2, dtype=dtype, batch=batch)
        g = synth.features(n, K, seed=3, dtype=dtype, batch=batch)
        for use_row in (True, False):
            got = nat.spmm_value_bw(row.to(dev) if use_row else None, rp.to(dev), c.to(dev), x.to(dev),
                                    g.to(dev), reduce)
            exact = oc.spmm_value_bw(oc.F64, reduce, row.numpy(), rp.numpy(), c.numpy(),
                                     x.double().numpy(), g.double().numpy())
            l1 = oc.spmm_value_bw(oc.F64, reduce, row.numpy(), rp.numpy(), c.numpy(),
                                  x.double().abs().numpy(), g.double().abs().numpy())
            err = np.abs(got.cpu().double().numpy() - exact)
            assert (err <= SUM_TOL[dtype] * l1 + SUM_ATOL[dtype]).all()


@pytest.mark.parametrize('dtype', FLOAT_DTYPES)
@pytest.mark.parametrize('reduce', ['min', 'max'])
def test_minmax_bw(dev, dtype, reduce):
    rp, c = synth.rmat_csr(9, 10, seed=6)
    n, E = 1 << 9, c.numel()
    for K, batch, has_value in ((16, (), True), (5, (2, ), True), (32, (), False)):
        v, x = make_inputs(rp, c, n, K, dtype, has_value, batch)
        gout = synth.features(n, K, seed=9, dtype=dtype, batch=batch)
        out, arg = run_gpu(dev, rp, c, v, x, reduce)
        gv, gm = nat.spmm_minmax_bw(rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev),
                                    gout.to(dev), arg, want_value=has_value, want_mat=True)
        egv, egm = oc.spmm_minmax_bw(oc.F64, c.numpy(), None if v is None else v.double().numpy(),
                                     x.double().numpy(), gout.double().numpy(), arg.cpu().numpy(),
                                     want_value=has_value)
        tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 4e-3,
               torch.bfloat16: 3e-2}[dtype]
        # grad_mat: hardware atomics in the element type (like the reference's scatter_add_, one
        # rounding per addition, any order): |err| <= (#addends + 1) * u * sum|terms|
        absv = None if v is None else v.double().abs().numpy()
        _, l1 = oc.spmm_minmax_bw(oc.F64, c.numpy(), absv, x.double().numpy(), gout.double().abs().numpy(),
                                  arg.cpu().numpy(), want_value=False)
        _, cnt = oc.spmm_minmax_bw(oc.F64, c.numpy(), None, x.double().numpy(), np.ones_like(gout.double().numpy()),
                                   arg.cpu().numpy(), want_value=False)
        u = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -11,
             torch.bfloat16: 2.0 ** -8}[dtype]
        err = np.abs(gm.cpu().double().numpy() - egm)
        floor = 2.0 ** -25 if dtype == torch.float16 else 1e-40  # half a subnormal step per addition
        assert (err <= (cnt + 1) * (u * l1 * 1.01 + floor)).all(), float((err / ((cnt + 1) * u * l1 + 1e-30)).max())
        if has_value:
            scale = max(1.0, float(np.abs(egv).max()))
            assert np.allclose(gv.cpu().double().numpy(), egv, rtol=tol, atol=tol * scale)


def _csc_arrays(rp, c, n_cols):
    """(colptr, csr2csc, row) of a CSR pattern, by stable host argsort (independent of the product's sort)."""
    E = c.numel()
    row = torch.repeat_interleave(torch.arange(rp.numel() - 1), rp[1:] - rp[:-1])
    perm = torch.from_numpy(np.argsort((c * (rp.numel() - 1) + row).numpy(), kind='stable'))
    colptr = torch.zeros(n_cols + 1, dtype=torch.int64)
    colptr[1:] = torch.cumsum(torch.bincount(c, minlength=n_cols), 0)
    assert perm.numel() == E
    return colptr, perm, row


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float64])
def test_minmax_bw_winner_lists_hub_columns(dev, dtype):
    if not experiments_build():
        pytest.skip('the winner-list route is only compiled into experiment builds (scripts/variants.py)')
    _minmax_bw_winner_lists_hub_columns(dev, dtype)


def _minmax_bw_winner_lists_hub_columns(dev, dtype):
    """The winner-list route of tsamd_spmm_minmax_bw_csc (csrc/spmm_bw_list.hip, TSAMD_MINMAX_BW_LISTS=1, K <= 1024) on a
    power-law graph whose hub columns span many 256-position waves of the pull kernel (head / tail carries + fix-up)
    and whose long rows span many 64-entry chunks of the list kernel (offsets of cut rows): against the default route
    (win masks + masked merge-path SpMM; same arithmetic, other summation order) and against
    exact integer arithmetic (small integer operands: every sum is exact in fp32, so BOTH routes must agree bit for
    bit); K up to the tile limit, a batch, value-less, columns without entries."""
    import os
    rp, c = synth.rmat_csr(12, 24, seed=11)
    n, E = 1 << 12, c.numel()
    deg_col = torch.bincount(c, minlength=n)
    assert int(deg_col.max()) > 1024 and int((deg_col == 0).sum()) > 0
    colptr, perm, row = _csc_arrays(rp, c, n)
    g = torch.Generator().manual_seed(3)
    for K, batch, has_value in ((128, (), False), (96, (2, ), True), (1024, (), True), (300, (), False)):
        shape = tuple(batch) + (n, K)
        x = torch.randint(-8, 9, shape, generator=g).to(dtype)
        gout = torch.randint(-4, 5, shape, generator=g).to(dtype)
        v = torch.randint(1, 4, (E, ), generator=g).to(dtype) if has_value else None
        out, arg = run_gpu(dev, rp, c, v, x, 'max')
        args = (rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev), gout.to(dev), arg,
                colptr.to(dev), perm.to(dev), row.to(dev))
        _, gm_masks = nat.spmm_minmax_bw_csc(*args, want_value=False, want_mat=True)
        os.environ['TSAMD_MINMAX_BW_LISTS'] = '1'
        try:
            _, gm = nat.spmm_minmax_bw_csc(*args, want_value=False, want_mat=True)
        finally:
            del os.environ['TSAMD_MINMAX_BW_LISTS']
        _, gm_scatter = nat.spmm_minmax_bw(rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev), gout.to(dev),
                                           arg, want_value=False, want_mat=True)
        if dtype == torch.bfloat16:  # sums beyond 256 round: the two pulls (fp32 sums, one rounding) still agree exactly
            assert bits_equal(gm, gm_masks), (K, batch)
        else:
            assert bits_equal(gm, gm_masks) and bits_equal(gm, gm_scatter), (K, batch)
        # exact reference on the host: scatter of the integer products
        a = arg.cpu()
        valid = a != E
        a0 = a.masked_fill(~valid, 0)
        term = gout.double() * (v.double()[a0] if has_value else 1.0)
        term = term.masked_fill(~valid, 0)
        want = torch.zeros(shape, dtype=torch.float64)
        if batch:
            for b in range(batch[0]):
                want[b].scatter_add_(0, c[a0[b]], term[b])
        else:
            want.scatter_add_(0, c[a0], term)
        if dtype != torch.bfloat16:
            assert torch.equal(gm.cpu().double(), want), (K, batch)
        else:
            assert torch.equal(gm.cpu(), want.to(dtype)), (K, batch)  # exact sum, rounded once


@pytest.fixture(params=['masks', 'lists'])
def pull_route(request):
    """Both grad_mat routes of tsamd_spmm_minmax_bw_csc: the default (win masks + masked merge-path SpMM) and the
    winner lists (TSAMD_MINMAX_BW_LISTS=1)."""
    import os
    if request.param == 'lists':
        if not experiments_build():
            pytest.skip('the winner-list route is only compiled into experiment builds (scripts/variants.py)')
        os.environ['TSAMD_MINMAX_BW_LISTS'] = '1'
    yield request.param
    os.environ.pop('TSAMD_MINMAX_BW_LISTS', None)


@pytest.mark.parametrize('reduce', ['min', 'max'])
@pytest.mark.parametrize('dtype', FLOAT_DTYPES)
def test_minmax_bw_csc_pull(dev, dtype, reduce, pull_route):
    """tsamd_spmm_minmax_bw_csc (winner masks + masked merge-path SpMM over the CSC view) against the fp64
    formulas of csrc/spmm.cpp:204-242: fp32 (fp64) accumulation, one rounding -- so well inside the bound of the
    scatter kernel -- deterministic, identical grad_value; rows above and below 64 entries, K that is not a
    multiple of 32 / 64, batches, value-less."""
    rp, c = synth.rmat_csr(9, 10, seed=6)
    n, E = 1 << 9, c.numel()
    assert int((rp[1:] - rp[:-1]).max()) > 64  # the long-row mask path is exercised
    colptr, perm, row = _csc_arrays(rp, c, n)
    for K, batch, has_value in ((16, (), True), (5, (2, ), True), (32, (), False), (70, (), True), (128, (2, ), False),
                                (192, (), True)):
        v, x = make_inputs(rp, c, n, K, dtype, has_value, batch)
        gout = synth.features(n, K, seed=9, dtype=dtype, batch=batch)
        out, arg = run_gpu(dev, rp, c, v, x, reduce)
        args = (rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev), gout.to(dev), arg,
                colptr.to(dev), perm.to(dev), row.to(dev))
        gv, gm = nat.spmm_minmax_bw_csc(*args, want_value=has_value, want_mat=True)
        gv2, gm2 = nat.spmm_minmax_bw_csc(*args, want_value=has_value, want_mat=True)
        assert bits_equal(gm, gm2), 'grad_mat is not deterministic'
        egv, egm = oc.spmm_minmax_bw(oc.F64, c.numpy(), None if v is None else v.double().numpy(),
                                     x.double().numpy(), gout.double().numpy(), arg.cpu().numpy(),
                                     want_value=has_value)
        absv = None if v is None else v.double().abs().numpy()
        _, l1 = oc.spmm_minmax_bw(oc.F64, c.numpy(), absv, x.double().numpy(), gout.double().abs().numpy(),
                                  arg.cpu().numpy(), want_value=False)
        u = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -11,
             torch.bfloat16: 2.0 ** -8}[dtype]
        # every product rounded to the element type (u * |term| each), summed in fp32 / fp64, rounded once
        err = np.abs(gm.cpu().double().numpy() - egm)
        floor = 2.0 ** -24 if dtype == torch.float16 else 1e-40
        _, cnt = oc.spmm_minmax_bw(oc.F64, c.numpy(), None, x.double().numpy(), np.ones_like(gout.double().numpy()),
                                   arg.cpu().numpy(), want_value=False)
        acc_u = 2.0 ** -53 if dtype == torch.float64 else 2.0 ** -24
        bound = u * l1 * 1.01 + u * np.abs(egm) * 1.01 + (cnt + 1) * acc_u * l1 + floor
        assert (err <= bound).all(), (K, batch, float((err / (bound + 1e-300)).max()))
        if has_value:  # masked SDDMM over the records (16-byte rows) or the row-parallel LDS kernel: fp32 sums, one rounding
            assert bits_equal(gv, gv2), 'grad_value is not deterministic'
            tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
            scale = max(1.0, float(np.abs(egv).max()))
            assert np.allclose(gv.cpu().double().numpy(), egv, rtol=tol, atol=tol * scale), K
        # through autograd: the front-end hands the CSC arrays over when `mat` needs a gradient
        if batch == ():
            import pytorch_sparse_amd as ts
            xr = x.to(dev).requires_grad_()
            vr = None if v is None else v.to(dev).requires_grad_()
            A = ts.SparseTensor(rowptr=rp.to(dev), col=c.to(dev), value=vr, sparse_sizes=(n, n), is_sorted=True,
                                trust_data=True)
            # the front-end takes the pull for rows of >= 64 features (>= 128 for 2-byte rows with grad_value) and
            # whenever reproducible gradients are asked for -- BEFORE the forward, which is when it builds and hands
            # over the CSC arrays (test_minmax_backward_route_rule pins the rule itself)
            torch.use_deterministic_algorithms(True)
            try:
                o = A.matmul(xr, reduce)
                o.backward(gout.to(dev))
            finally:
                torch.use_deterministic_algorithms(False)
            assert bits_equal(xr.grad, gm), 'autograd path differs from the C-ABI call'
            assert A.storage.has_csr2csc() and A.storage.has_colptr()
            if has_value:
                assert bits_equal(vr.grad, gv)


@pytest.mark.parametrize('dtype', FLOAT_DTYPES)
def test_minmax_arg32_forward_and_pull(dev, dtype):
    """tsamd_spmm_minmax_arg32 / tsamd_spmm_minmax_bw_csc_arg32 (the ids SparseTensor.matmul keeps for its own backward):
    the int32 ids equal the API's int64 arg_out, `out` is bit-identical, and the pull on them gives the same bits as on
    the int64 ids; every packet width of the forward (K = 1, 2, 6, 64, 128), cut rows, batches, empty rows."""
    rp, c = synth.rmat_csr(10, 16, seed=2)  # max degree > 128-item partitions => cut rows (fix-up kernel)
    n, E = 1 << 10, c.numel()
    colptr, perm, row = _csc_arrays(rp, c, n)
    for reduce in ('min', 'max'):
        for K, batch, has_value in ((1, (), True), (2, (), False), (6, (2, ), True), (64, (), True), (128, (), False),
                                    (128, (2, ), True)):
            v, x = make_inputs(rp, c, n, K, dtype, has_value, batch)
            vd = None if v is None else v.to(dev)
            out, arg = run_gpu(dev, rp, c, v, x, reduce)
            out32, arg32 = nat.spmm_minmax_arg32(rp.to(dev), c.to(dev), vd, x.to(dev), reduce)
            assert arg32.dtype == torch.int32 and bits_equal(out, out32), (reduce, K)
            assert torch.equal(arg32.long(), arg), (reduce, K)
            gout = synth.features(n, K, seed=9, dtype=dtype, batch=batch).to(dev)
            sddmm = has_value and (K * x.element_size()) % 16 == 0
            a = (rp.to(dev), c.to(dev), vd, x.to(dev), gout)
            b = (colptr.to(dev), perm.to(dev), row.to(dev))
            gv, gm = nat.spmm_minmax_bw_csc(*a, arg, *b, want_value=sddmm, want_mat=True)
            gv32, gm32 = nat.spmm_minmax_bw_csc(*a, arg32, *b, want_value=sddmm, want_mat=True)
            assert bits_equal(gm, gm32), (reduce, K)
            if sddmm:
                assert bits_equal(gv, gv32), (reduce, K)
    # rows that are not 16-byte packets + grad_value: the int32 entry says so (the torch glue then widens the ids)
    v, x = make_inputs(rp, c, n, 6, dtype, True, ())
    out32, arg32 = nat.spmm_minmax_arg32(rp.to(dev), c.to(dev), v.to(dev), x.to(dev), 'max')
    gout = synth.features(n, 6, seed=9, dtype=dtype).to(dev)
    if (6 * x.element_size()) % 16 != 0:
        with pytest.raises(nat.TsamdError):
            nat.spmm_minmax_bw_csc(rp.to(dev), c.to(dev), v.to(dev), x.to(dev), gout, arg32, colptr.to(dev),
                                   perm.to(dev), row.to(dev), want_value=True, want_mat=True)


@pytest.mark.parametrize('dtype', FLOAT_DTYPES)
def test_masked_sddmm_pipelined_against_round4_kernel(dev, dtype):
    if not experiments_build():
        pytest.skip('TSAMD_MASKED_SDDMM_PIPE is only read by experiment builds (scripts/variants.py)')
    _masked_sddmm_pipelined_against_round4_kernel(dev, dtype)


def _masked_sddmm_pipelined_against_round4_kernel(dev, dtype):
    """grad_value of the pull backward: the pipelined masked SDDMM (record words of 8 steps in one round trip, the
    gathers of two steps in flight, v_dot2c for 2-byte types) against the round-4 kernel (TSAMD_MASKED_SDDMM_PIPE=0)
    for every lane-group width (1 ... 64 packets per row, also counts that are not powers of two), batches, a chunk
    that ends inside a wave (E % 64 != 0), and against the fp64 formula."""
    import os
    rp, c = synth.rmat_csr(10, 12, seed=4)
    n, E = 1 << 10, c.numel()
    assert E % 64 != 0
    colptr, perm, row = _csc_arrays(rp, c, n)
    vec = 16 // torch.empty(0, dtype=dtype).element_size()
    u = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
    for slots, batch in ((1, ()), (2, (2, )), (3, ()), (4, ()), (8, ()), (16, ()), (24, ()), (32, (2, )), (64, ())):
        K = slots * vec
        v, x = make_inputs(rp, c, n, K, dtype, True, batch)
        gout = synth.features(n, K, seed=9, dtype=dtype, batch=batch)
        out, arg = run_gpu(dev, rp, c, v, x, 'max')
        args = (rp.to(dev), c.to(dev), v.to(dev), x.to(dev), gout.to(dev), arg, colptr.to(dev), perm.to(dev), row.to(dev))
        gv, gm = nat.spmm_minmax_bw_csc(*args, want_value=True, want_mat=True)
        os.environ['TSAMD_MASKED_SDDMM_PIPE'] = '0'
        try:
            gv_old, gm_old = nat.spmm_minmax_bw_csc(*args, want_value=True, want_mat=True)
        finally:
            os.environ.pop('TSAMD_MASKED_SDDMM_PIPE', None)
        assert bits_equal(gm, gm_old)
        egv, _ = oc.spmm_minmax_bw(oc.F64, c.numpy(), v.double().numpy(), x.double().numpy(), gout.double().numpy(),
                                   arg.cpu().numpy(), want_value=True)
        l1, _ = oc.spmm_minmax_bw(oc.F64, c.numpy(), v.double().numpy(), x.double().abs().numpy(),
                                  gout.double().abs().numpy(), arg.cpu().numpy(), want_value=True)
        acc_u = 2.0 ** -53 if dtype == torch.float64 else 2.0 ** -24
        bound = (K * len(batch or (1, )) * 2 + 2) * acc_u * l1 + u * np.abs(egv) * 1.01 + 1e-30
        for got in (gv, gv_old):
            err = np.abs(got.cpu().double().numpy() - egv)
            assert (err <= bound).all(), (slots, batch, float((err / bound).max()))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_masked_sddmm_non_winners_contribute_nothing(dev, dtype):
    """An Inf (or NaN) of grad_out in a feature an entry did NOT win must not reach that entry's grad_value (the
    reference only ever touches the winner, spmm.cpp:222-231): the dot2 form masks BOTH operands."""
    K = 16
    rp = torch.tensor([0, 2])
    c = torch.tensor([0, 1])
    x = torch.ones(2, K, dtype=dtype)
    x[0, 1::2] = 9.0   # entry 0 wins the odd features, entry 1 the even ones
    x[1, 0::2] = 5.0
    g = torch.full((1, K), 0.5, dtype=dtype)
    g[0, 0] = float('inf')
    g[0, 2] = float('nan')
    v = torch.tensor([1.0, 1.0], dtype=dtype)
    colptr, perm, row = _csc_arrays(rp, c, 2)
    out, arg = nat.spmm(rp.to(dev), c.to(dev), v.to(dev), x.to(dev), 'max')
    assert arg[0, 0].item() == 1 and arg[0, 1].item() == 0
    gv, gm = nat.spmm_minmax_bw_csc(rp.to(dev), c.to(dev), v.to(dev), x.to(dev), g.to(dev), arg, colptr.to(dev),
                                    perm.to(dev), row.to(dev), want_value=True, want_mat=True)
    gv = gv.float().cpu()
    assert torch.isfinite(gv[0]) and abs(float(gv[0]) - 9.0 * 0.5 * (K // 2)) < 0.26, gv  # 8 odd features
    assert not torch.isfinite(gv[1])  # entry 1 did win the Inf / NaN features


def test_minmax_backward_route_rule(dev):
    """Which backward SparseTensor.matmul(x, 'max') prepares (pytorch_sparse_amd/tensor.py: storage_spmm, table in
    profiles/r05_minmax_bw_route_rule.md): the pull needs the CSC arrays, so a storage that has them after the forward
    took it.  Either way the gradient stays within the scatter kernel's bound of the fp64 formulas."""
    import pytorch_sparse_amd as ts
    rp, c = synth.rmat_csr(9, 10, seed=6)
    n = 1 << 9
    for dtype, K, with_value, det, want_pull in ((torch.float32, 32, False, False, False),
                                                 (torch.float32, 64, False, False, True),
                                                 (torch.bfloat16, 64, False, False, True),
                                                 (torch.bfloat16, 64, True, False, False),
                                                 (torch.bfloat16, 128, True, False, True),
                                                 (torch.float32, 64, True, False, True),
                                                 (torch.float32, 16, True, True, True)):
        v, x = make_inputs(rp, c, n, K, dtype, True)
        xr = x.to(dev).requires_grad_()
        vr = v.to(dev).requires_grad_(with_value)
        A = ts.SparseTensor(rowptr=rp.to(dev), col=c.to(dev), value=vr, sparse_sizes=(n, n), is_sorted=True,
                            trust_data=True)
        gout = synth.features(n, K, seed=9, dtype=dtype)
        torch.use_deterministic_algorithms(det)
        try:
            o = A.matmul(xr, 'max')
            assert A.storage.has_csr2csc() == want_pull, (dtype, K, with_value, det)
            o.backward(gout.to(dev))
        finally:
            torch.use_deterministic_algorithms(False)
        arg = nat.spmm(rp.to(dev), c.to(dev), v.to(dev), x.to(dev), 'max')[1]
        egv, egm = oc.spmm_minmax_bw(oc.F64, c.numpy(), v.double().numpy(), x.double().numpy(), gout.double().numpy(),
                                     arg.cpu().numpy(), want_value=True)
        _, l1 = oc.spmm_minmax_bw(oc.F64, c.numpy(), v.double().abs().numpy(), x.double().numpy(),
                                  gout.double().abs().numpy(), arg.cpu().numpy(), want_value=False)
        _, cnt = oc.spmm_minmax_bw(oc.F64, c.numpy(), None, x.double().numpy(), np.ones_like(gout.double().numpy()),
                                   arg.cpu().numpy(), want_value=False)
        u = 2.0 ** -24 if dtype == torch.float32 else 2.0 ** -8
        bound = (cnt + 2) * u * l1 * 1.01 + 1e-30
        err = np.abs(xr.grad.cpu().double().numpy() - egm)
        assert (err <= bound).all(), (dtype, K, float((err / bound).max()))
        if with_value:
            tol = 1e-5 if dtype == torch.float32 else 3e-2
            assert np.allclose(vr.grad.cpu().double().numpy(), egv, rtol=tol, atol=tol * max(1.0, float(np.abs(egv).max())))


def test_minmax_bw_csc_no_winner_and_empty(dev):
    """Rows without entries and elements without a winner (arg == E) contribute nothing; columns without
    entries get zeros (every element of grad_mat is written)."""
    rp = torch.tensor([0, 3, 3, 5, 5])
    c = torch.tensor([0, 1, 2, 0, 0])
    x = torch.tensor([[1., 5.], [1., 7.], [float('nan'), 7.], [0., 0.]])
    g = torch.tensor([[1., 2.], [3., 4.], [5., 6.], [7., 8.]])
    colptr, perm, row = _csc_arrays(rp, c, 4)
    out, arg = nat.spmm(rp.to(dev), c.to(dev), None, x.to(dev), 'max')
    gv, gm = nat.spmm_minmax_bw_csc(rp.to(dev), c.to(dev), None, x.to(dev), g.to(dev), arg, colptr.to(dev), perm.to(dev),
                                    row.to(dev), want_value=False, want_mat=True)
    _, gm_ref = nat.spmm_minmax_bw(rp.to(dev), c.to(dev), None, x.to(dev), g.to(dev), arg, want_value=False, want_mat=True)
    assert torch.equal(gm.cpu(), gm_ref.cpu())
    assert torch.equal(gm.cpu()[3], torch.zeros(2))


def test_ind2ptr_ptr2ind(dev):
    assert nat.ind2ptr(torch.tensor([2, 2, 4, 5, 5, 6], device=dev), 8).tolist() == \
        [0, 0, 0, 2, 2, 3, 5, 6, 6]
    assert nat.ptr2ind(torch.tensor([0, 0, 0, 2, 2, 3, 5, 6, 6], device=dev), 6).tolist() == \
        [2, 2, 4, 5, 5, 6]
    assert nat.ind2ptr(torch.zeros(0, dtype=torch.int64, device=dev), 4).tolist() == [0] * 5
    rp, c = synth.rmat_csr(14, 20, seed=2)
    row = nat.ptr2ind(rp.to(dev), c.numel())
    assert np.array_equal(row.cpu().numpy(), oc.ptr2ind(rp.numpy(), c.numel()))
    back = nat.ind2ptr(row, 1 << 14)
    assert torch.equal(back.cpu(), rp)
    # skewed laws: hub rows spread over many workgroups, long runs of empty rows (both kernels have a
    # separate path for them), entries only in the first / last row
    rng = np.random.RandomState(3)
    for M, ind in ((3_000_000, np.sort(rng.randint(0, 3_000_000, 4000))),          # tiles span > 2048 rows
                   (2_000_000, np.full(100_000, 1_999_999)),                        # all in the last row
                   (2_000_000, np.zeros(100_000, np.int64)),                        # all in the first row
                   (5000, np.sort(np.concatenate([np.full(3_000_000, 777), rng.randint(0, 5000, 20_000)]))),
                   (1_500_000, np.sort(np.concatenate([rng.randint(0, 10, 50_000),
                                                       rng.randint(1_400_000, 1_500_000, 50_000)])))):
        ind = ind.astype(np.int64)
        want = oc.ind2ptr(ind, M)
        got = nat.ind2ptr(torch.from_numpy(ind).to(dev), M)
        assert np.array_equal(got.cpu().numpy(), want)
        assert np.array_equal(nat.ptr2ind(got, ind.size).cpu().numpy(), ind)


def test_determinism(dev):
    rp, c = synth.rmat_csr(14, 20, seed=0, device=dev)
    v = synth.values(c.numel(), device=dev)
    x = synth.features(1 << 14, 128, device=dev)
    a, _ = nat.spmm(rp, c, v, x, 'sum')
    b, _ = nat.spmm(rp, c, v, x, 'sum')
    assert torch.equal(a, b)


@pytest.mark.parametrize('config', ['c2', 'ns'])
def test_full_size_properties(dev, config):
    """BASELINE.json sizes: exact checks that do not need the (slow) oracle.
    Small-integer inputs make every fp32 sum exact, so
      * column checksum: sum_m out[m,:] == sum_c weight[c] * x[c,:]     (checksum of checksums)
      * max: out == value[arg] * x[col[arg]] and arg lies in the row's edge range,
        and a sampled set of rows is checked in full against the oracle."""
    scale, K = (20, 64) if config == 'c2' else (21, 128)
    n = 1 << scale
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
    E = c.numel()
    g = torch.Generator(device=dev).manual_seed(4)
    v = torch.randint(1, 4, (E, ), generator=g, device=dev).float()
    x = torch.randint(-4, 5, (n, K), generator=g, device=dev).float()
    out, _ = nat.spmm(rp, c, v, x, 'sum')
    w = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, c, v.double())
    expect = (w[:, None] * x.double()).sum(0)
    assert torch.equal(out.double().sum(0), expect)
    # sampled rows (incl. the heaviest) against the oracle, bit-exact because sums are exact
    deg = rp[1:] - rp[:-1]
    rows = torch.cat([torch.topk(deg, 4).indices.cpu(), torch.randint(0, n, (60, ))]).unique()
    rpc, cc, vc, xc = rp.cpu(), c.cpu(), v.cpu(), x.cpu()
    for reduce in ('sum', 'max'):
        o, a = nat.spmm(rp, c, v, x, reduce)
        for r in rows.tolist():
            s, e = int(rpc[r]), int(rpc[r + 1])
            eo, ea = oracle_spmm(torch.tensor([0, e - s]), cc[s:e], vc[s:e], xc, reduce)
            assert torch.equal(o[r].cpu(), eo[0]), (reduce, r)
            if ea is not None:
                ea = torch.where(ea == e - s, torch.full_like(ea, E), ea + s)
                assert torch.equal(a[r].cpu(), ea[0]), (reduce, r)
        if a is not None:
            valid = a != E
            assert torch.equal(valid.any(1), deg > 0)
            ai = torch.where(valid, a, torch.zeros_like(a))
            prod = v[ai] * x[c[ai], torch.arange(K, device=dev)[None, :]]
            assert torch.equal(torch.where(valid, prod, torch.zeros_like(prod)), o)
            assert ((ai >= rp[:-1, None]) & (ai < rp[1:, None]) | ~valid).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float64])
def test_relabel_path_is_bit_identical(dev, dtype, monkeypatch):
    """The channel-camping avoidance (hashed-row copy of mat) only changes addresses:
    forced on and forced off must give bit-identical results, batched input included, and the
    N that is not a power of two exercises the cycle-walking hash."""
    n, m, K = 9001, 5000, 8
    g = torch.Generator().manual_seed(11)
    E = 1300000
    row, col = torch.randint(0, m, (E, ), generator=g), torch.randint(0, n, (E, ), generator=g)
    col[::3] &= ~7  # low-bit skew, as the probe looks for
    rp, c = synth.to_csr(row, col, m, n)
    v, x = make_inputs(rp, c, n, K, dtype, True, batch=(2, ))
    res = {}
    if not experiments_build():
        pytest.skip('TSAMD_SPMM_RELABEL is only read by experiment builds; tests/test_relabelled_gpu.py covers the copy')
    for mode in ('0', '1', 'auto'):
        monkeypatch.setenv('TSAMD_SPMM_RELABEL', mode)
        for reduce in ('sum', 'max'):
            out, arg = run_gpu(dev, rp, c, v, x, reduce)
            res[(mode, reduce)] = (out.cpu(), None if arg is None else arg.cpu())
    for reduce in ('sum', 'max'):
        for mode in ('1', 'auto'):
            assert bits_equal(res[('0', reduce)][0], res[(mode, reduce)][0]), (mode, reduce)
        if reduce == 'max':
            assert torch.equal(res[('0', reduce)][1], res[('1', reduce)][1])
    check_spmm(res[('1', 'max')][0], res[('1', 'max')][1], rp, c, v, x, 'max')
    check_spmm(res[('1', 'sum')][0], None, rp, c, v, x, 'sum')


def test_spmm_fuzz_small_shapes(dev):
    """300 random small problems (ragged shapes, empty leading/trailing rows, rows ending exactly on
    partition boundaries, K not a multiple of anything, batch dims) against the oracle."""
    rng = np.random.RandomState(1234)
    dts = [torch.float32, torch.float64, torch.bfloat16, torch.float16, torch.int32, torch.int64]
    for case in range(300):
        M = int(rng.choice([1, 2, 3, 7, 64, 129, 500, 1025]))
        N = int(rng.choice([1, 2, 5, 33, 128, 1000]))
        kind = rng.randint(4)
        if kind == 0:      # uniform random degrees
            deg = rng.randint(0, 6, size=M)
        elif kind == 1:    # exactly 127/128/129 items per stretch: partition edges on row ends
            deg = rng.choice([0, 63, 64, 65, 127, 128], size=M)
        elif kind == 2:    # one heavy row, everything else empty
            deg = np.zeros(M, dtype=np.int64)
            deg[rng.randint(M)] = int(rng.choice([1, 64, 1000, 5000]))
        else:              # heavy head, empty tail
            deg = np.where(np.arange(M) < max(1, M // 4), rng.randint(0, 300, size=M), 0)
        rp = torch.zeros(M + 1, dtype=torch.int64)
        rp[1:] = torch.from_numpy(np.cumsum(deg))
        E = int(rp[-1])
        c = torch.from_numpy(rng.randint(0, N, size=E)).long()
        dtype = dts[rng.randint(len(dts))]
        K = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 33, 64, 100, 130]))
        batch = () if rng.rand() < 0.7 else (int(rng.randint(1, 4)), )
        has_value = bool(rng.rand() < 0.6)
        reduce = ['sum', 'mean', 'min', 'max'][rng.randint(4)]
        v, x = make_inputs(rp, c, N, K, dtype, has_value, batch, seed=case)
        out, arg = run_gpu(dev, rp, c, v, x, reduce)
        try:
            check_spmm(out, arg, rp, c, v, x, reduce)
        except AssertionError as exc:
            raise AssertionError('case %d: M=%d N=%d E=%d K=%d %s %s batch=%s value=%s kind=%d: %s' % (
                case, M, N, E, K, dtype, reduce, batch, has_value, kind, exc))


def test_backward_kernels_full_size_exact(dev):
    """BASELINE size (2^20 R-MAT, F = 64): small-integer inputs make every product and sum exact in
    fp32, so the value gradient and the min/max backward can be compared bit-for-bit with the
    reference's own formulas evaluated by ATen in fp64 (csrc/spmm.cpp:96-98, 204-242)."""
    scale, K = 20, 64
    n = 1 << scale
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
    E = c.numel()
    row = nat.ptr2ind(rp, E)
    g = torch.Generator(device=dev).manual_seed(8)
    x = torch.randint(-3, 4, (n, K), generator=g, device=dev).float()
    go = torch.randint(-3, 4, (n, K), generator=g, device=dev).float()
    v = torch.randint(1, 4, (E, ), generator=g, device=dev).float()
    # value gradient: sum over k of x[col] * g[row], chunked to bound the temporary
    gv = nat.spmm_value_bw(row, rp, c, x, go, 'sum')
    for s in range(0, E, 1 << 22):
        e = min(E, s + (1 << 22))
        ref = (x[c[s:e]].double() * go[row[s:e]].double()).sum(1)
        assert torch.equal(gv[s:e].double(), ref), s
    deg = (rp[1:] - rp[:-1]).clamp(min=1)
    gvm = nat.spmm_value_bw(None, rp, c, x, go, 'mean')
    assert torch.allclose(gvm, gv / deg[row].float(), rtol=1e-6, atol=0)
    # min/max backward against the ATen composition of the reference
    out, arg = nat.spmm(rp, c, v, x, 'max')
    gval, gmat = nat.spmm_minmax_bw(rp, c, v, x, go, arg)
    invalid = arg == E
    a = arg.masked_fill(invalid, 0)
    ind = c[a]
    contrib = (x.gather(0, ind) * go).masked_fill(invalid, 0)
    ref_gval = torch.zeros(E, dtype=torch.float64, device=dev).scatter_add_(0, a.flatten(), contrib.double().flatten())
    ref_gmat = torch.zeros(n, K, dtype=torch.float64, device=dev).scatter_add_(0, ind, (v[a] * go).masked_fill(invalid, 0).double())
    assert torch.equal(gval.double(), ref_gval)
    assert torch.equal(gmat.double(), ref_gmat)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float64, torch.uint8])
def test_gather_rows_matches_index_select(dev, dtype):
    """tsamd_gather_rows (pack step of the sharded SpMM's row exchange): every packet width (16 / 8 / 4 / 2 / 1
    bytes by pitch), duplicates, negative ids."""
    import pytorch_sparse_amd  # noqa: F401
    g = torch.Generator().manual_seed(2)
    for n, k in ((1000, 128), (777, 6), (50, 3), (4096, 1), (300, 7), (10, 33)):
        src = (torch.rand(n, k, generator=g) * 200).to(dtype).to(dev)
        idx = torch.randint(-n, n, (2 * n + 3, ), generator=g).to(dev)
        got = torch.ops.tsamd.gather_rows(src, idx)
        assert bits_equal(got, src[idx])
    assert torch.ops.tsamd.gather_rows(src, idx[:0]).shape == (0, 33)


@pytest.mark.parametrize('reduce', ['sum', 'max'])
def test_spmm_permuted_equals_materialised_view(dev, reduce):
    """tsamd_spmm_permuted(colptr, row, value, csr2csc) == tsamd_spmm on the materialised CSC arrays, bit for
    bit (the entries are the same, in the same order)."""
    import ctypes
    rp, c = synth.rmat_csr(11, 10, seed=8, device=dev)
    n, E = 1 << 11, c.numel()
    v = synth.values(E, device=dev)
    row = nat.ptr2ind(rp, E)
    perm = torch.argsort(c * n + row)
    colptr = nat.ind2ptr(c[perm].contiguous(), n)
    for K in (64, 12):
        x = synth.features(n, K, seed=4, device=dev)
        want, warg = nat.spmm(colptr, row[perm].contiguous(), v[perm].contiguous(), x, reduce)
        out = torch.empty_like(want)
        arg = torch.empty_like(warg) if warg is not None else None
        L = nat.lib()
        red = nat.REDUCES[reduce]
        nb = L.tsamd_spmm_workspace_bytes(0, red, ctypes.c_int64(1), ctypes.c_int64(n), ctypes.c_int64(n),
                                          ctypes.c_int64(K), ctypes.c_int64(E))
        ws = nat.workspace(nb, x.device)
        p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
        st = L.tsamd_spmm_permuted(0, red, p(colptr), p(row), p(v), p(perm), p(x), p(out), p(arg), ctypes.c_int64(1),
                                   ctypes.c_int64(n), ctypes.c_int64(n), ctypes.c_int64(K), ctypes.c_int64(E), p(ws),
                                   ctypes.c_size_t(ws.numel()), nat.stream_ptr(x.device))
        assert st == 0
        torch.cuda.synchronize()
        assert bits_equal(out, want)
        if arg is not None:
            assert torch.equal(arg, warg)


def test_small_int_mean_with_wrapped_divisor_is_pinned(dev):
    """Documented divergence (csrc/common.h mean_of): the reference divides the wrapped sum by the row length
    CAST TO THE ELEMENT TYPE (reducer.h:74), which is 0 -- a SIGFPE there -- for a uint8 / int8 row of 256
    entries or an int16 row of 65536.  Here such a row gives 0; lengths that wrap to a non-zero divisor follow
    the reference's arithmetic exactly (300 entries of uint8: divisor 300 & 255 = 44)."""
    for dtype, wrap in ((torch.uint8, 256), (torch.int8, 256), (torch.int16, 65536)):
        for deg in (wrap, 2 * wrap, wrap + 44):
            rp = torch.tensor([0, deg, deg + 3])
            c = torch.cat([torch.arange(deg) % 7, torch.tensor([0, 1, 2])])
            x = torch.arange(1, 8, dtype=torch.int64).view(7, 1).repeat(1, 4).to(dtype)
            out, _ = nat.spmm(rp.to(dev), c.to(dev), None, x.to(dev), 'mean')
            s = int(x[c[:deg], 0].to(torch.int64).sum())
            info = torch.iinfo(dtype)
            span = info.max - info.min + 1
            wrapped_sum = (s - info.min) % span + info.min
            div = (deg - info.min) % span + info.min if dtype != torch.uint8 else deg % 256
            if dtype != torch.uint8:
                div = (deg + (1 << (info.bits - 1))) % span - (1 << (info.bits - 1))
            want0 = 0 if div == 0 else int(wrapped_sum / div)  # C++ integer division truncates toward zero
            assert out[0].cpu().tolist() == [want0] * 4, (dtype, deg, out[0].cpu().tolist(), want0)
            assert out[1].cpu().tolist() == [2] * 4  # (1 + 2 + 3) / 3: the short row next to it is untouched


def test_minmax_all_init_value_reports_no_winner(dev):
    """A row whose every candidate equals the reducer's init value (or is NaN) has no strict winner
    (reducer.h:63-67): out = the init value as the reference writes it, arg = E here (the reference leaves
    a stale index -- the one documented divergence of arg_out, masked in tests/test_oracle.py)."""
    big = torch.finfo(torch.float32).max
    rp = torch.tensor([0, 2, 4])
    c = torch.tensor([0, 1, 0, 1])
    x = torch.tensor([[big, float('nan')], [big, float('nan')]])
    out, arg = nat.spmm(rp.to(dev), c.to(dev), None, x.to(dev), 'min')
    assert arg.cpu().tolist() == [[4, 4], [4, 4]]
    assert out.cpu()[:, 0].tolist() == [big, big]
    lo = torch.tensor([[-big, float('nan')], [-big, float('nan')]])
    out, arg = nat.spmm(rp.to(dev), c.to(dev), None, lo.to(dev), 'max')
    assert arg.cpu().tolist() == [[4, 4], [4, 4]] and out.cpu()[:, 0].tolist() == [-big, -big]


@pytest.mark.parametrize('dtype', ALL_DTYPES)
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
def test_reference_order_mode_is_bit_identical_to_the_reference(dev, dtype, reduce):
    """tsamd_spmm_reference_order(1): the forward in the reference CPU kernel's order of operations (csrc/cpu/spmm_cpu.cpp:
    61-87: entries of a row one after the other, product and sum rounded separately in the element type) -- EVERY bit of
    the output equals the compiled reference's (oracle/_ref), fp sums included; hub rows (thousands of terms), empty rows,
    batches, value-less matrices, through the C-ABI and through the drop-in torch op."""
    from tests.baseline_configs import ref_spmm_cpu
    rp, c = synth.rmat_csr(11, 24, seed=2)  # max degree in the thousands
    n = 1 << 11
    try:
        assert torch.ops.tsamd.reference_order(1) == 1
        for K, has_value, batch in ((32, True, ()), (5, False, ()), (16, True, (2, ))):
            v, x = make_inputs(rp, c, n, K, dtype, has_value, batch, seed=3)
            want, warg, kind = ref_spmm_cpu(rp, c, v, x, reduce)
            out, arg = run_gpu(dev, rp, c, v, x, reduce)
            if kind == 'reference':  # (without oracle/_ref the C restatement accumulates 2-byte types in fp32)
                assert bits_equal(out.cpu(), want), (K, has_value, batch)
            else:
                check_spmm(out, arg, rp, c, v, x, reduce)
            if reduce in ('min', 'max'):
                # rows with entries whose result beat the reducer's initial value (elsewhere the reference leaves an
                # unset index, DESIGN.md section 5: an integer row of 255s under uint8 min)
                info = torch.finfo(dtype) if dtype.is_floating_point else torch.iinfo(dtype)
                init = info.max if reduce == 'min' else info.min
                live = (rp[1:] > rp[:-1]).view(-1, 1).expand(n, K) & (want != init)
                assert torch.equal(arg.cpu()[live], warg[live])
        if dtype == torch.float32:  # the drop-in op takes the same switch
            import pytorch_sparse_amd as ts
            v, x = make_inputs(rp, c, n, 32, dtype, True, (), seed=4)
            A = ts.SparseTensor(rowptr=rp.to(dev), col=c.to(dev), value=v.to(dev), sparse_sizes=(n, n), is_sorted=True,
                                trust_data=True)
            want = ref_spmm_cpu(rp, c, v, x, reduce)[0]
            got = A.matmul(x.to(dev), reduce=reduce)
            assert bits_equal(got.cpu(), want)
    finally:
        assert torch.ops.tsamd.reference_order(0) == 0
    # back on the product kernels: same values within the documented bound
    v, x = make_inputs(rp, c, n, 32, dtype, True, (), seed=3)
    out, arg = run_gpu(dev, rp, c, v, x, reduce)
    check_spmm(out, arg, rp, c, v, x, reduce)

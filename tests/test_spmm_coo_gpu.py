"""torch_sparse.spmm(index, value, m, n, matrix) on small unsorted COO inputs: the one-launch route
(tsamd_spmm_coo_small, csrc/spmm_coo.hip) against the semantics of torch_sparse/spmm.py:25-31 -- index_select,
multiply, scatter_add -- evaluated with ATen in fp64 / int64 on the host, and against the sorted route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64, torch.uint8, torch.int8,
          torch.int16]


def _inputs(E, m, n, K, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    index = torch.stack([torch.randint(0, m, (E, ), generator=g), torch.randint(0, n, (E, ), generator=g)])
    if E > 10:
        index[:, E // 2] = index[:, 1]  # a duplicated (row, col) pair: the two entries add up
    if dtype.is_floating_point:
        value = (torch.rand(E, generator=g) * 2 - 1).to(dtype)
        x = (torch.rand(n, K, generator=g) * 2 - 1).to(dtype)
    else:
        lo, hi = (0, 6) if dtype == torch.uint8 else (-5, 6)
        value = torch.randint(lo, hi, (E, ), generator=g).to(dtype)
        x = torch.randint(lo, hi, (n, K), generator=g).to(dtype)
    return index, value, x


def _expected(index, value, x, m):
    """torch_sparse/spmm.py:25-31 in wide arithmetic -> (sum, L1 mass), both [m, K] float64 / int64."""
    wide = torch.float64 if value.dtype.is_floating_point else torch.int64
    prod = x.to(wide).index_select(0, index[1]) * value.to(wide).unsqueeze(-1)
    if value.dtype in (torch.float16, torch.bfloat16):
        prod = prod.to(value.dtype).to(wide)  # the reference multiplies in the element type
    out = torch.zeros(m, x.size(1), dtype=wide).index_add_(0, index[0], prod)
    l1 = torch.zeros(m, x.size(1), dtype=wide).index_add_(0, index[0], prod.abs())
    return out, l1


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('E,m,n,K', [(5000, 1000, 1000, 16), (1, 3, 2, 5), (0, 4, 4, 8), (777, 50, 4000, 33),
                                     (20000, 9000, 300, 8), (4096, 1, 10, 128), (300, 5000, 5000, 64)])
def test_small_coo_matches_reference_semantics(dev, dtype, E, m, n, K):
    import pytorch_sparse_amd as ts
    index, value, x = _inputs(E, m, n, K, dtype, seed=E + K)
    assert torch.ops.tsamd.spmm_coo_small_supported(value.to(dev), E, m, K)
    out = ts.spmm(index.to(dev), value.to(dev), m, n, x.to(dev))
    assert out.dtype == dtype and tuple(out.shape) == (m, K)
    want, l1 = _expected(index, value, x, m)
    if dtype.is_floating_point:
        tol = {torch.float32: 1e-5, torch.float64: 1e-13, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
        err = (out.cpu().double() - want).abs()
        assert bool((err <= tol * l1 + 1e-30).all()), float((err / l1.clamp(min=1e-30)).max())
    else:  # exact, wrapping like the element type
        assert torch.equal(out.cpu(), want.to(dtype))


def test_small_coo_agrees_with_sorted_route_and_with_fixture(dev):
    """The same call with a gradient requested takes the sorted CSR route: both must agree (integers exactly), and the
    configs[0] fixture written by the reference Python is met by the one-launch route."""
    import os
    import pytorch_sparse_amd as ts
    for dtype in (torch.float32, torch.int64):
        index, value, x = _inputs(3000, 400, 500, 24, dtype, seed=9)
        a = ts.spmm(index.to(dev), value.to(dev), 400, 500, x.to(dev))
        rowptr_route = ts.SparseTensor(row=index[0].to(dev), col=index[1].to(dev), value=value.to(dev),
                                       sparse_sizes=(400, 500)).matmul(x.to(dev))
        if dtype == torch.int64:
            assert torch.equal(a, rowptr_route)
        else:
            assert torch.allclose(a, rowptr_route, rtol=1e-5, atol=1e-5)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'py7_c1_spmm.npz'))
    out = ts.spmm(torch.from_numpy(z['index']).to(dev), torch.from_numpy(z['value']).to(dev), int(z['m']), int(z['n']),
                  torch.from_numpy(z['mat']).to(dev))
    d = (out.cpu().double() - torch.from_numpy(z['out']).double()).abs()
    assert bool((d <= 1e-5 * torch.from_numpy(z['l1']).double() + 1e-30).all())


def test_small_coo_route_is_one_launch_and_skipped_under_autograd(dev):
    import pytorch_sparse_amd as ts
    index, value, x = _inputs(2000, 300, 300, 16, torch.float32, seed=1)
    index, value, x = index.to(dev), value.to(dev), x.to(dev)
    ts.spmm(index, value, 300, 300, x)
    torch.cuda.synchronize()
    prev = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode('error')  # no host sync anywhere in the call
    try:
        out = ts.spmm(index, value, 300, 300, x)
    finally:
        torch.cuda.set_sync_debug_mode(prev)
    # gradients: the differentiable route, same values
    v2, x2 = value.clone().requires_grad_(), x.clone().requires_grad_()
    out2 = ts.spmm(index, v2, 300, 300, x2)
    out2.sum().backward()
    assert torch.allclose(out, out2.detach(), rtol=1e-5, atol=1e-5)
    assert v2.grad is not None and x2.grad is not None
    # too big for the direct route: falls through to the sorted one
    assert not torch.ops.tsamd.spmm_coo_small_supported(value, 1 << 20, 300, 16)


def test_small_coo_skips_ids_out_of_range_and_honours_deterministic_mode(dev):
    """a column id outside [0, N) must not read past `mat` (the entry is skipped, like a row id outside
    [0, M)); under torch.use_deterministic_algorithms(True) the call takes the sorted route, whose sums are reproducible
    (the one-launch route adds through LDS atomics in arrival order)."""
    import pytorch_sparse_amd as ts
    E, m, n, K = 3000, 200, 150, 16
    index, value, x = _inputs(E, m, n, K, torch.float32, seed=9)
    bad = index.clone()
    bad[1, 7] = n            # one past the last column
    bad[1, 1234] = -3        # negative
    bad[1, 2999] = 1 << 40   # far out
    keep = torch.ones(E, dtype=torch.bool)
    keep[[7, 1234, 2999]] = False
    out = ts.spmm(bad.to(dev), value.to(dev), m, n, x.to(dev))
    want, l1 = _expected(index[:, keep], value[keep], x, m)
    err = (out.cpu().double() - want).abs()
    assert bool((err <= 1e-5 * l1 + 1e-30).all())
    # deterministic algorithms: bit-identical run to run, and equal to the sorted route
    index_d, value_d, x_d = index.to(dev), value.to(dev), x.to(dev)
    was = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        a = ts.spmm(index_d, value_d, m, n, x_d)
        b = ts.spmm(index_d, value_d, m, n, x_d)
    finally:
        torch.use_deterministic_algorithms(was)
    assert torch.equal(a, b)
    v2 = value_d.clone().requires_grad_()  # (autograd: the sorted route as well)
    assert torch.equal(a, ts.spmm(index_d, v2, m, n, x_d).detach())

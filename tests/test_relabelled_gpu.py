"""The relabelled ("channel-camping free") layout must be the plain path bit for bit -- only the
addresses change (pytorch_sparse_amd/relabelled.py, include/tsamd.h tsamd_spmm_relabelled)."""
import pytest
import torch

from tests.util import bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ts():
    import pytorch_sparse_amd
    return pytorch_sparse_amd


def _graph(ts, dev, scale, ef, dtype, with_value, rect=False):
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(scale, ef, seed=3, device=dev)
    n = 1 << scale
    if rect:  # fewer rows than columns: the two position maps differ
        m = n // 2 + 17
        rp = rp[:m + 1].clone()
        c = c[:int(rp[-1])]
    else:
        m = n
    v = synth.values(c.numel(), dtype=dtype, device=dev) if with_value else None
    return ts.SparseTensor(rowptr=rp, col=c, value=v, sparse_sizes=(m, n), is_sorted=True, trust_data=True), m, n


def test_relabel_index_is_a_bijection(ts, dev):
    for n in (1, 2, 3, 64, 1000, 4097, 1 << 17):
        h = ts.relabel_index(n, dev)
        assert h.numel() == n and torch.equal(torch.sort(h).values, torch.arange(n, device=dev))
        x = torch.arange(n, device=dev, dtype=torch.float32).view(n, 1).repeat(1, 3)
        assert torch.equal(ts.from_relabelled(ts.to_relabelled(x)), x)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float64])
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
def test_forward_is_bit_identical(ts, dev, dtype, reduce):
    from pytorch_sparse_amd import synth
    for with_value, rect, K, batch in ((True, False, 128, ()), (False, True, 20, ()), (True, True, 7, (2, ))):
        A, m, n = _graph(ts, dev, 12, 16, dtype, with_value, rect)
        x = synth.features(n, K, dtype=dtype, device=dev, batch=batch)
        plain = ts.matmul(A, x, reduce)
        out_h, arg_h = ts.matmul_relabelled(A, ts.to_relabelled(x), reduce, return_arg=True)
        assert bits_equal(ts.from_relabelled(out_h), plain)
        if reduce in ('min', 'max'):
            rowptr, col, value = A.csr()
            _, arg = (torch.ops.torch_sparse.spmm_min if reduce == 'min' else torch.ops.torch_sparse.spmm_max)(
                rowptr, col, value, x)
            assert torch.equal(ts.from_relabelled(arg_h), arg)


@pytest.mark.parametrize('reduce', ['sum', 'mean'])
def test_gradients_are_bit_identical(ts, dev, reduce):
    from pytorch_sparse_amd import synth
    A0, m, n = _graph(ts, dev, 12, 16, torch.float32, True, rect=True)
    rowptr, col, v0 = A0.csr()
    x0 = synth.features(n, 48, device=dev)
    g = synth.features(m, 48, seed=9, device=dev)
    res = []
    for relabelled in (False, True):
        v = v0.clone().requires_grad_()
        x = x0.clone().requires_grad_()
        A = ts.SparseTensor(rowptr=rowptr, col=col, value=v, sparse_sizes=(m, n), is_sorted=True, trust_data=True)
        if relabelled:
            out = ts.from_relabelled(ts.matmul_relabelled(A, ts.to_relabelled(x), reduce))
        else:
            out = ts.matmul(A, x, reduce)
        out.backward(g)
        res.append((out.detach(), v.grad, x.grad))
    for a, b in zip(*res):
        assert bits_equal(a, b)


def test_minmax_training_is_refused_in_this_layout(ts, dev):
    A, m, n = _graph(ts, dev, 8, 8, torch.float32, True)
    x = torch.randn(n, 8, device=dev, requires_grad=True)
    with pytest.raises(RuntimeError, match='no backward'):
        ts.matmul_relabelled(A, x, 'max')


def test_two_layers_without_leaving_the_layout(ts, dev):
    """out_h of one product is the operand of the next (square matrix): P (A relu(A X))."""
    from pytorch_sparse_amd import synth
    A, m, n = _graph(ts, dev, 13, 12, torch.float32, True)
    x = synth.features(n, 64, device=dev)
    plain = ts.matmul(A, torch.relu(ts.matmul(A, x)))
    y_h = ts.matmul_relabelled(A, torch.relu(ts.matmul_relabelled(A, ts.to_relabelled(x))))
    assert bits_equal(ts.from_relabelled(y_h), plain)

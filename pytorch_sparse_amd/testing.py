"""Parametrisation helpers mirroring torch_sparse/testing.py (GPU only: there is no CPU path)."""
import torch

reductions = ['sum', 'add', 'mean', 'min', 'max']
dtypes = [torch.half, torch.bfloat16, torch.float, torch.double, torch.int, torch.long]
grad_dtypes = [torch.half, torch.bfloat16, torch.float, torch.double]
devices = [torch.device('cuda:0')] if torch.cuda.is_available() else []


def tensor(x, dtype, device):
    return None if x is None else torch.tensor(x, dtype=dtype, device=device)


import torch as _torch
from torch import Tensor as _Tensor

from .matmul import matmul as _matmul
from .tensor import SparseTensor as _SparseTensor


class Aggregation(_torch.nn.Module):
    """``matmul(adj_t, x, reduce)`` as a module: what a jittable message-passing layer does with its
    adjacency.  ``torch.jit.script(Aggregation('mean'))`` is the one-line check that the TorchScript
    boundary works (used by ``__graft_entry__.smoke()``)."""

    def __init__(self, reduce: str = 'sum'):
        super().__init__()
        self.reduce = reduce

    def forward(self, adj_t: _SparseTensor, x: _Tensor) -> _Tensor:
        return _matmul(adj_t, x, self.reduce)

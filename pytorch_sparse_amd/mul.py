"""Sparse * dense-vector broadcasts (reference: torch_sparse/mul.py:22-40, add.py:21-36):
``mul(adj, d.view(-1, 1))`` scales rows, ``mul(adj, d.view(1, -1))`` scales columns -- the other
half of ``D^-1/2 A D^-1/2``.  The pattern (and every cache) is shared with the input; only the
value tensor is new.  Sparse*sparse and add() with sparse operands stay out of scope."""
import torch
from torch import Tensor

from .tensor import SparseTensor


def _broadcast(src: SparseTensor, other: Tensor) -> Tensor:
    """other expanded to one entry per stored element (row-wise [M,1,...] or column-wise [1,N,...])."""
    if other.dim() >= 2 and other.size(0) == src.size(0) and other.size(1) == 1:
        return other.squeeze(1)[src.storage.row()]
    if other.dim() >= 2 and other.size(0) == 1 and other.size(1) == src.size(1):
        return other.squeeze(0)[src.storage.col()]
    raise ValueError('Size mismatch: Expected size (%d, 1, ...) or (1, %d, ...), but got size %s.'
                     % (src.size(0), src.size(1), tuple(other.size())))


def mul(src: SparseTensor, other: Tensor) -> SparseTensor:
    if not isinstance(other, Tensor):
        raise NotImplementedError('sparse * sparse is outside the accelerated hot path')
    other = _broadcast(src, other)
    value = src.storage.value()
    value = other if value is None else other.to(value.dtype) * value
    return src.set_value(value, layout='coo')


def mul_nnz(src: SparseTensor, other: Tensor, layout=None) -> SparseTensor:
    value = src.storage.value()
    return src.set_value(other if value is None else value * other, layout=layout or 'coo')


def add(src: SparseTensor, other: Tensor) -> SparseTensor:
    if not isinstance(other, Tensor):
        raise NotImplementedError('sparse + sparse is outside the accelerated hot path')
    other = _broadcast(src, other)
    value = src.storage.value()
    value = other + 1 if value is None else other.to(value.dtype) + value
    return src.set_value(value, layout='coo')


SparseTensor.mul = lambda self, other: mul(self, other)
SparseTensor.mul_nnz = lambda self, other, layout=None: mul_nnz(self, other, layout)
SparseTensor.add = lambda self, other: add(self, other)
SparseTensor.__mul__ = SparseTensor.mul
SparseTensor.__add__ = SparseTensor.add

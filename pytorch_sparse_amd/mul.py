"""Element-wise ``mul`` / ``add`` of a SparseTensor with a broadcast dense vector or another
SparseTensor (reference: torch_sparse/mul.py, torch_sparse/add.py).

``mul(adj, d.view(-1, 1))`` scales rows, ``mul(adj, d.view(1, -1))`` scales columns -- the other half
of ``D^-1/2 A D^-1/2``.  With a dense operand the pattern (and every cache) is
shared with the input; only the value tensor is new.  Sparse (x) sparse goes through the package's
radix sort: ``add`` = concatenate + sort + ``coalesce('sum')`` (add.py:41-59), ``mul`` = concatenate +
stable sort + keep the adjacent equal pairs (mul.py:43-80).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .tensor import SparseTensor


def _broadcast(src: SparseTensor, other: Tensor) -> Tensor:
    """other expanded to one entry per stored element (row-wise [M,1,...] or column-wise [1,N,...])."""
    if other.dim() >= 2 and other.size(0) == src.size(0) and other.size(1) == 1:
        return other.squeeze(1).index_select(0, src.storage.row())
    if other.dim() >= 2 and other.size(0) == 1 and other.size(1) == src.size(1):
        return other.squeeze(0).index_select(0, src.storage.col())
    raise ValueError('Size mismatch: Expected size (%d, 1, ...) or (1, %d, ...), but got size (%d, %d, ...).'
                     % (src.size(0), src.size(1), other.size(0), other.size(1) if other.dim() > 1 else 1))


def _concat(src: SparseTensor, other: SparseTensor) -> Tuple[Tensor, Tensor, Optional[Tensor], Tuple[int, int]]:
    rowA, colA, valueA = src.coo()
    rowB, colB, valueB = other.coo()
    row = torch.cat([rowA, rowB], dim=0)
    col = torch.cat([colA, colB], dim=0)
    value: Optional[Tensor] = None
    if valueA is not None and valueB is not None:
        value = torch.cat([valueA, valueB], dim=0)
    sizes = (max(src.size(0), other.size(0)), max(src.size(1), other.size(1)))
    return row, col, value, sizes


@torch.jit._overload  # noqa: F811
def mul(src, other):  # noqa: F811
    # type: (SparseTensor, Tensor) -> SparseTensor
    pass


@torch.jit._overload  # noqa: F811
def mul(src, other):  # noqa: F811
    # type: (SparseTensor, SparseTensor) -> SparseTensor
    pass


def mul(src, other):  # noqa: F811
    if isinstance(other, Tensor):
        bc = _broadcast(src, other)
        value = src.storage.value()
        if value is not None:
            bc = bc.to(value.dtype) * value
        return src.set_value(bc, layout='coo')

    assert isinstance(other, SparseTensor)
    if not src.is_coalesced():
        raise ValueError("The `src` tensor is not coalesced")
    if not other.is_coalesced():
        raise ValueError("The `other` tensor is not coalesced")
    row, col, value, sizes = _concat(src, other)
    if value is None:
        raise ValueError('Both sparse tensors must contain values')
    # stable sort: of two equal keys the entry of `src` comes first, the entry of `other` second
    row, col, perm = torch.ops.tsamd.sort_coo(row, col, sizes[0], sizes[1], True)
    second = torch.zeros(row.numel(), dtype=torch.bool, device=row.device)
    if row.numel() > 1:
        second[1:] = (row[1:] == row[:-1]) & (col[1:] == col[:-1])
    row, col, pos, _ = torch.ops.tsamd.filter_coo('mask', row, col, second, 0, 0, False, 0, 0, True, True)
    value = value.index_select(0, perm.index_select(0, pos - 1)) * \
        value.index_select(0, perm.index_select(0, pos))
    return SparseTensor(row=row, rowptr=None, col=col, value=value, sparse_sizes=sizes, is_sorted=True,
                        trust_data=False)


def mul_(src: SparseTensor, other: Tensor) -> SparseTensor:
    other = _broadcast(src, other)
    value = src.storage.value()
    value = other if value is None else value.mul_(other.to(value.dtype))
    return src.set_value_(value, layout='coo')


def mul_nnz(src: SparseTensor, other: Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = other if value is None else value.mul(other.to(value.dtype))
    return src.set_value(value, layout=layout)


def mul_nnz_(src: SparseTensor, other: Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = other if value is None else value.mul_(other.to(value.dtype))
    return src.set_value_(value, layout=layout)


@torch.jit._overload  # noqa: F811
def add(src, other):  # noqa: F811
    # type: (SparseTensor, Tensor) -> SparseTensor
    pass


@torch.jit._overload  # noqa: F811
def add(src, other):  # noqa: F811
    # type: (SparseTensor, SparseTensor) -> SparseTensor
    pass


def add(src, other):  # noqa: F811
    if isinstance(other, Tensor):
        bc = _broadcast(src, other)
        value = src.storage.value()
        if value is None:
            bc = bc + 1
        else:
            bc = bc.to(value.dtype) + value
        return src.set_value(bc, layout='coo')
    elif isinstance(other, SparseTensor):
        row, col, value, sizes = _concat(src, other)
        out = SparseTensor(row=row, rowptr=None, col=col, value=value, sparse_sizes=sizes, is_sorted=False,
                           trust_data=False)
        return out.coalesce(reduce='sum')
    raise NotImplementedError


def add_(src: SparseTensor, other: Tensor) -> SparseTensor:
    other = _broadcast(src, other)
    value = src.storage.value()
    value = other + 1 if value is None else value.add_(other.to(value.dtype))
    return src.set_value_(value, layout='coo')


def add_nnz(src: SparseTensor, other: Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = other.add(1) if value is None else value.add(other.to(value.dtype))
    return src.set_value(value, layout=layout)


def add_nnz_(src: SparseTensor, other: Tensor, layout: Optional[str] = None) -> SparseTensor:
    value = src.storage.value()
    value = other.add(1) if value is None else value.add_(other.to(value.dtype))
    return src.set_value_(value, layout=layout)


SparseTensor.mul = lambda self, other: mul(self, other)
SparseTensor.mul_ = lambda self, other: mul_(self, other)
SparseTensor.mul_nnz = lambda self, other, layout=None: mul_nnz(self, other, layout)
SparseTensor.mul_nnz_ = lambda self, other, layout=None: mul_nnz_(self, other, layout)
SparseTensor.__mul__ = SparseTensor.mul
SparseTensor.__rmul__ = SparseTensor.mul
SparseTensor.__imul__ = SparseTensor.mul_
SparseTensor.add = lambda self, other: add(self, other)
SparseTensor.add_ = lambda self, other: add_(self, other)
SparseTensor.add_nnz = lambda self, other, layout=None: add_nnz(self, other, layout)
SparseTensor.add_nnz_ = lambda self, other, layout=None: add_nnz_(self, other, layout)
SparseTensor.__add__ = SparseTensor.add
SparseTensor.__radd__ = SparseTensor.add
SparseTensor.__iadd__ = SparseTensor.add_

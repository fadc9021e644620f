"""SparseTensor @ dense (SpMM, four reductions, autograd) and SparseTensor @ SparseTensor (SpSpMM).

Front-ends of the reference (torch_sparse/matmul.py:9-171) over ``torch.ops.torch_sparse.spmm_*``
(argument order and optional-cache handling identical) and the fused ``tsamd::spspmm`` op.
"""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .tensor import SparseTensor, storage_spmm


def spmm_sum(src: SparseTensor, other: Tensor) -> Tensor:
    return storage_spmm(src.storage, other, 'sum')[0]


def spmm_add(src: SparseTensor, other: Tensor) -> Tensor:
    return spmm_sum(src, other)


def spmm_mean(src: SparseTensor, other: Tensor) -> Tensor:
    return storage_spmm(src.storage, other, 'mean')[0]


def spmm_min(src: SparseTensor, other: Tensor) -> Tuple[Tensor, Tensor]:
    out, arg = storage_spmm(src.storage, other, 'min')
    assert arg is not None
    return out, arg


def spmm_max(src: SparseTensor, other: Tensor) -> Tuple[Tensor, Tensor]:
    out, arg = storage_spmm(src.storage, other, 'max')
    assert arg is not None
    return out, arg


def spmm(src: SparseTensor, other: Tensor, reduce: str = 'sum') -> Tensor:
    return storage_spmm(src.storage, other, reduce, False)[0]


def spspmm_sum(src: SparseTensor, other: SparseTensor) -> SparseTensor:
    """C = A * B, rows of C sorted by column and coalesced (what the reference gets from
    torch.sparse.mm, matmul.py:94-111).  The result carries values iff an input does."""
    assert src.sparse_size(1) == other.sparse_size(0)
    rowptrA, colA, valueA = src.csr()
    rowptrB, colB, valueB = other.csr()
    with_value = valueA is not None or valueB is not None
    if valueA is not None and valueB is not None and valueA.dtype != valueB.dtype:
        dtype = torch.promote_types(valueA.dtype, valueB.dtype)
        valueA, valueB = valueA.to(dtype), valueB.to(dtype)
    if valueA is not None:
        valueA = valueA.detach()
    if valueB is not None:
        valueB = valueB.detach()
    rowptrC, colC, valueC = torch.ops.tsamd.spspmm(rowptrA, colA, valueA, rowptrB, colB, valueB,
                                                   other.sparse_size(1), with_value)
    value: Optional[Tensor] = None
    if with_value:
        value = valueC
    return SparseTensor(row=None, rowptr=rowptrC, col=colC, value=value,
                        sparse_sizes=(src.sparse_size(0), other.sparse_size(1)), is_sorted=True,
                        trust_data=True)


def spspmm_add(src: SparseTensor, other: SparseTensor) -> SparseTensor:
    return spspmm_sum(src, other)


def spspmm(src: SparseTensor, other: SparseTensor, reduce: str = 'sum') -> SparseTensor:
    if reduce == 'sum' or reduce == 'add':
        return spspmm_sum(src, other)
    if reduce == 'mean' or reduce == 'min' or reduce == 'max':
        raise NotImplementedError
    raise ValueError


@torch.jit._overload  # noqa: F811
def matmul(src, other, reduce):  # noqa: F811
    # type: (SparseTensor, Tensor, str) -> Tensor
    pass


@torch.jit._overload  # noqa: F811
def matmul(src, other, reduce):  # noqa: F811
    # type: (SparseTensor, SparseTensor, str) -> SparseTensor
    pass


def matmul(src, other, reduce='sum'):  # noqa: F811
    """Sparse @ dense -> dense (reduce in sum|add|mean|min|max); sparse @ sparse -> sparse (sum).
    Scriptable through the two overloads above (the reference's idiom, matmul.py:129-171)."""
    if isinstance(other, Tensor):
        return spmm(src, other, reduce)
    elif isinstance(other, SparseTensor):
        return spspmm(src, other, reduce)
    raise ValueError


SparseTensor.spmm = lambda self, other, reduce='sum': spmm(self, other, reduce)
SparseTensor.spspmm = lambda self, other, reduce='sum': spspmm(self, other, reduce)
SparseTensor.matmul = lambda self, other, reduce='sum': matmul(self, other, reduce)
SparseTensor.__matmul__ = lambda self, other: matmul(self, other, 'sum')

"""SparseTensor @ dense (SpMM, four reductions, autograd) and SparseTensor @ SparseTensor (SpSpMM).

Front-ends of the reference (torch_sparse/matmul.py:9-171) over ``torch.ops.torch_sparse.spmm_*``
(argument order and optional-cache handling identical) and the fused ``tsamd::spspmm`` op.
"""
from typing import Tuple, Union

import torch
from torch import Tensor

from .tensor import SparseTensor


def _spmm_args(src: SparseTensor, other: Tensor, need_rowcount: bool):
    """Collect the cached arrays the op needs; fill the CSC-side caches only when a gradient
    w.r.t. `other` will be asked for (reference matmul.py:12-25, 38-53)."""
    rowptr, col, value = src.csr()
    st = src.storage
    row, csr2csc, colptr, rowcount = st._row, st._csr2csc, st._colptr, st._rowcount
    if value is not None:
        value = value.to(other.dtype)
        if value.requires_grad:
            row = st.row()
    if other.requires_grad:
        row, csr2csc, colptr = st.row(), st.csr2csc(), st.colptr()
        if need_rowcount:
            rowcount = st.rowcount()
    return row, rowptr, col, value, rowcount, colptr, csr2csc


def spmm_sum(src: SparseTensor, other: Tensor) -> Tensor:
    row, rowptr, col, value, _, colptr, csr2csc = _spmm_args(src, other, False)
    return torch.ops.torch_sparse.spmm_sum(row, rowptr, col, value, colptr, csr2csc, other)


def spmm_add(src: SparseTensor, other: Tensor) -> Tensor:
    return spmm_sum(src, other)


def spmm_mean(src: SparseTensor, other: Tensor) -> Tensor:
    row, rowptr, col, value, rowcount, colptr, csr2csc = _spmm_args(src, other, True)
    return torch.ops.torch_sparse.spmm_mean(row, rowptr, col, value, rowcount, colptr, csr2csc,
                                            other)


def _spmm_minmax(src: SparseTensor, other: Tensor, op) -> Tuple[Tensor, Tensor]:
    rowptr, col, value = src.csr()
    if value is not None:
        value = value.to(other.dtype)
    return op(rowptr, col, value, other)


def spmm_min(src: SparseTensor, other: Tensor) -> Tuple[Tensor, Tensor]:
    return _spmm_minmax(src, other, torch.ops.torch_sparse.spmm_min)


def spmm_max(src: SparseTensor, other: Tensor) -> Tuple[Tensor, Tensor]:
    return _spmm_minmax(src, other, torch.ops.torch_sparse.spmm_max)


def spmm(src: SparseTensor, other: Tensor, reduce: str = 'sum') -> Tensor:
    if reduce in ('sum', 'add'):
        return spmm_sum(src, other)
    if reduce == 'mean':
        return spmm_mean(src, other)
    if reduce == 'min':
        return spmm_min(src, other)[0]
    if reduce == 'max':
        return spmm_max(src, other)[0]
    raise ValueError


def spspmm_sum(src: SparseTensor, other: SparseTensor) -> SparseTensor:
    """C = A * B, rows of C sorted by column and coalesced (what the reference gets from
    torch.sparse.mm, matmul.py:94-111).  The result carries values iff an input does."""
    assert src.sparse_size(1) == other.sparse_size(0)
    rowptrA, colA, valueA = src.csr()
    rowptrB, colB, valueB = other.csr()
    with_value = valueA is not None or valueB is not None
    if with_value:
        dtype = valueA.dtype if valueA is not None else valueB.dtype
        if valueB is not None and valueA is not None:
            dtype = torch.promote_types(valueA.dtype, valueB.dtype)
        valueA = None if valueA is None else valueA.detach().to(dtype)
        valueB = None if valueB is None else valueB.detach().to(dtype)
    rowptrC, colC, valueC = torch.ops.tsamd.spspmm(rowptrA, colA, valueA, rowptrB, colB, valueB,
                                                   other.sparse_size(1), with_value)
    return SparseTensor(rowptr=rowptrC, col=colC, value=valueC if with_value else None,
                        sparse_sizes=(src.sparse_size(0), other.sparse_size(1)), is_sorted=True,
                        trust_data=True)


def spspmm_add(src: SparseTensor, other: SparseTensor) -> SparseTensor:
    return spspmm_sum(src, other)


def spspmm(src: SparseTensor, other: SparseTensor, reduce: str = 'sum') -> SparseTensor:
    if reduce in ('sum', 'add'):
        return spspmm_sum(src, other)
    if reduce in ('mean', 'min', 'max'):
        raise NotImplementedError
    raise ValueError


def matmul(src: SparseTensor, other: Union[Tensor, SparseTensor], reduce: str = 'sum'):
    """Sparse @ dense -> dense (reduce in sum|add|mean|min|max); sparse @ sparse -> sparse (sum)."""
    if isinstance(other, Tensor):
        return spmm(src, other, reduce)
    if isinstance(other, SparseTensor):
        return spspmm(src, other, reduce)
    raise ValueError


SparseTensor.spmm = lambda self, other, reduce='sum': spmm(self, other, reduce)
SparseTensor.spspmm = lambda self, other, reduce='sum': spspmm(self, other, reduce)
SparseTensor.matmul = lambda self, other, reduce='sum': matmul(self, other, reduce)
SparseTensor.__matmul__ = lambda self, other: matmul(self, other, 'sum')

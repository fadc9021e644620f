"""Functional SpSpMM on raw COO tensors (API of torch_sparse/spspmm.py:6-33)."""
import torch

from .coalesce import sorted_unique


def _csr(index, value, rows, cols, sort):
    row, col = index[0], index[1]
    if sort:  # `coalesced=True` in the reference means "sort the inputs first"
        row, col, perm, _, _ = sorted_unique(row, col, rows, cols)
        if perm is not None and value is not None:
            value = value.index_select(0, perm)
    return torch.ops.torch_sparse.ind2ptr(row, rows), col, value


def spspmm(indexA, valueA, indexB, valueB, m, k, n, coalesced=False):
    """C = A (m x k) * B (k x n) for COO operands.  The operands have to be sorted row-major with
    unique entries; ``coalesced=True`` sorts them first.  Returns (indexC [2, nnzC], valueC) with
    the rows of C sorted by column and duplicates summed."""
    rowptrA, colA, valueA = _csr(indexA, valueA, m, k, coalesced)
    rowptrB, colB, valueB = _csr(indexB, valueB, k, n, coalesced)
    with_value = valueA is not None or valueB is not None
    if with_value and valueA is not None and valueB is not None and valueA.dtype != valueB.dtype:
        dtype = torch.promote_types(valueA.dtype, valueB.dtype)
        valueA, valueB = valueA.to(dtype), valueB.to(dtype)
    rowptrC, colC, valueC = torch.ops.tsamd.spspmm(
        rowptrA, colA, None if valueA is None else valueA.detach(), rowptrB, colB,
        None if valueB is None else valueB.detach(), n, with_value)
    rowC = torch.ops.torch_sparse.ptr2ind(rowptrC, colC.numel())
    return torch.stack([rowC, colC], dim=0), (valueC if with_value else None)

"""Concatenation of SparseTensors (reference: torch_sparse/cat.py).

``cat(ts, 0)`` stacks rows -- the inverse of the row partition ``narrow(0, ...)`` used by the sharded SpMM
 -- ``cat(ts, 1)`` stacks columns, ``cat(ts, (0, 1))`` stacks diagonally (mini-batching
of graphs), ``cat(ts, d >= 2)`` concatenates the dense value dimensions.  Row and diagonal stacking
only offset and append the index arrays (device memcpys).  Column stacking must interleave the
operands row by row: the reference concatenates and re-sorts (cat.py:160-165, ``is_sorted=False``);
here each operand is written straight to its final slots by one ``tsamd::scatter_rows`` launch --
no sort, no host sync.
"""
from typing import List, Optional

import torch
from torch import Tensor

from .storage import SparseStorage
from .tensor import SparseTensor


def _all(parts: List[Optional[Tensor]]) -> Optional[Tensor]:
    if any(p is None for p in parts):
        return None
    return torch.cat(parts, dim=0)


def _stack_ptrs(ptrs: List[Optional[Tensor]], nnzs: List[int]) -> Optional[Tensor]:
    if any(p is None for p in ptrs):
        return None
    out, off = [], 0
    for i, (p, n) in enumerate(zip(ptrs, nnzs)):
        out.append(p if i == 0 else p[1:] + off)
        off += n
    return torch.cat(out, dim=0)


def cat(tensors, dim):
    assert len(tensors) > 0
    if isinstance(dim, int):
        dim = tensors[0].dim() + dim if dim < 0 else dim
        if dim == 0:
            return cat_first(tensors)
        if dim == 1:
            return cat_second(tensors)
        if 1 < dim < tensors[0].dim():
            values = []
            for tensor in tensors:
                value = tensor.storage.value()
                assert value is not None
                values.append(value)
            return tensors[0].set_value(torch.cat(values, dim=dim - 1), layout='coo')
        raise IndexError('Dimension out of range: Expected to be in range of [%d, %d], but got %d.'
                         % (-tensors[0].dim(), tensors[0].dim() - 1, dim))
    assert isinstance(dim, (tuple, list))
    assert len(dim) == 2
    assert sorted(dim) == [0, 1]
    return cat_diag(tensors)


def cat_first(tensors: List[SparseTensor]) -> SparseTensor:
    """Reference cat.py:60-114."""
    sts = [t.storage for t in tensors]
    nnzs = [st.col().numel() for st in sts]
    if any(st._row is None for st in sts) and any(st._rowptr is None for st in sts):
        for st in sts:  # mixed COO-only / CSR-only operands: settle on CSR
            st.rowptr()
    M, rows = 0, []
    for st in sts:
        rows.append(None if st._row is None else st._row + M)
        M += st.sparse_size(0)
    N = max(st.sparse_size(1) for st in sts)
    storage = SparseStorage(row=_all(rows), rowptr=_stack_ptrs([st._rowptr for st in sts], nnzs),
                            col=torch.cat([st.col() for st in sts], dim=0),
                            value=_all([st.value() for st in sts]), sparse_sizes=(M, N),
                            rowcount=_all([st._rowcount for st in sts]), is_sorted=True,
                            trust_data=True)
    return tensors[0].from_storage(storage)


def cat_second(tensors: List[SparseTensor]) -> SparseTensor:
    """Reference cat.py:117-165 (same result as its concatenate-and-sort: within a row the entries
    of operand t keep their order and follow those of operands < t)."""
    sts = [t.storage for t in tensors]
    nnzs = [st.col().numel() for st in sts]
    M = max(st.sparse_size(0) for st in sts)
    N = sum(st.sparse_size(1) for st in sts)
    col0 = sts[0].col()
    if not col0.is_cuda:
        raise RuntimeError('pytorch_sparse_amd has no CPU implementation: move the SparseTensors to the GPU')
    total = sum(nnzs)

    ptrs = []
    for st, n in zip(sts, nnzs):  # row pointers padded to the common number of rows
        p = st.rowptr()
        if p.numel() < M + 1:
            p = torch.cat([p, p.new_full((M + 1 - p.numel(), ), n)])
        ptrs.append(p)
    rowptr = ptrs[0].clone()
    for p in ptrs[1:]:
        rowptr += p
    row = col0.new_empty(total)
    col = col0.new_empty(total)
    src = col0.new_empty(total)
    nxt = rowptr[:-1].clone()  # next free slot of every output row
    col_off, src_off = 0, 0
    for st, p, n in zip(sts, ptrs, nnzs):
        torch.ops.tsamd.scatter_rows(st.row(), st.col(), nxt - p[:-1], col_off, src_off, row, col, src)
        nxt = nxt + (p[1:] - p[:-1])
        col_off += st.sparse_size(1)
        src_off += n

    value = _all([st.value() for st in sts])
    if value is not None:
        value = value.index_select(0, src)
    storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value, sparse_sizes=(M, N),
                            colptr=_stack_ptrs([st._colptr for st in sts], nnzs),
                            colcount=_all([st._colcount for st in sts]), is_sorted=True,
                            trust_data=True)
    return tensors[0].from_storage(storage)


def cat_diag(tensors: List[SparseTensor]) -> SparseTensor:
    """Reference cat.py:168-261."""
    assert len(tensors) > 0
    sts = [t.storage for t in tensors]
    nnzs = [st.col().numel() for st in sts]
    if any(st._row is None for st in sts) and any(st._rowptr is None for st in sts):
        for st in sts:
            st.rowptr()
    M = N = nnz = 0
    rows, cols, csr2cscs, csc2csrs = [], [], [], []
    for st, n in zip(sts, nnzs):
        rows.append(None if st._row is None else st._row + M)
        cols.append(st.col() + N)
        csr2cscs.append(None if st._csr2csc is None else st._csr2csc + nnz)
        csc2csrs.append(None if st._csc2csr is None else st._csc2csr + nnz)
        M += st.sparse_size(0)
        N += st.sparse_size(1)
        nnz += n
    storage = SparseStorage(row=_all(rows), rowptr=_stack_ptrs([st._rowptr for st in sts], nnzs),
                            col=torch.cat(cols, dim=0), value=_all([st.value() for st in sts]),
                            sparse_sizes=(M, N), rowcount=_all([st._rowcount for st in sts]),
                            colptr=_stack_ptrs([st._colptr for st in sts], nnzs),
                            colcount=_all([st._colcount for st in sts]), csr2csc=_all(csr2cscs),
                            csc2csr=_all(csc2csrs), is_sorted=True, trust_data=True)
    return tensors[0].from_storage(storage)

"""Sub-matrix extraction: ``narrow``, ``select``, ``index_select``, ``masked_select``, ``permute`` and
``SparseTensor.__getitem__`` (reference: torch_sparse/narrow.py, select.py, index_select.py,
masked_select.py, permute.py, tensor.py:624-671).

SURVEY.md section 8f rank 3 -- the callers either side of the row-sharded SpMM (``narrow(0, ...)`` IS
the 1-D row partition, ``index_select`` / ``masked_select`` cut the mini-batch sub-graphs).  Same names,
arguments and results as the reference; the index work runs in two fused HIP primitives
(``tsamd::select_segments``, ``tsamd::filter_coo``; csrc/select.hip) instead of chains of
``rowcount[idx]`` / ``cumsum`` / ``repeat_interleave`` / ``gather_csr`` / boolean-mask indexing.  Each
call costs exactly one host sync (the output size is data dependent).  Values are gathered with
``index_select`` so they stay differentiable, as in the reference.
"""
from typing import Any, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .storage import SparseStorage, get_layout
from .tensor import SparseTensor


def _take(value: Optional[Tensor], pos: Tensor) -> Optional[Tensor]:
    return None if value is None else value.index_select(0, pos)


def _gpu(src: SparseTensor):
    if not src.storage.col().is_cuda:
        raise RuntimeError('pytorch_sparse_amd has no CPU implementation: move the SparseTensor to the GPU')


def narrow(src: SparseTensor, dim: int, start: int, length: int) -> SparseTensor:
    if dim < 0:
        dim = src.dim() + dim
    if start < 0:
        start = src.size(dim) + start

    if dim == 0:  # views of the CSR arrays, reference narrow.py:15-42
        rowptr, col, value = src.csr()
        assert 0 <= start and length >= 0 and start + length <= src.sparse_size(0)
        # both ends of the slice in one device -> host copy
        lo, hi = torch.stack([rowptr[start], rowptr[start + length]]).tolist()
        rowptr = rowptr[start:start + length + 1] - lo
        row = src.storage._row
        if row is not None:
            row = row[lo:hi] - start
        col = col[lo:hi]
        if value is not None:
            value = value[lo:hi]
        rowcount = src.storage._rowcount
        if rowcount is not None:
            rowcount = rowcount[start:start + length]
        storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                                sparse_sizes=(length, src.sparse_size(1)), rowcount=rowcount,
                                is_sorted=True, trust_data=True)
        return src.from_storage(storage)

    if dim == 1:  # reference narrow.py:44-72: (col >= start) & (col < start + length) compaction
        _gpu(src)
        row, col, value = src.coo()
        assert 0 <= start and length >= 0 and start + length <= src.sparse_size(1)
        row, col, pos, _ = torch.ops.tsamd.filter_coo('col_range', row, col, None, start, length, False,
                                                      0, start, True, True)
        colptr = src.storage._colptr
        if colptr is not None:
            colptr = colptr[start:start + length + 1]
            colptr = colptr - colptr[0]
        colcount = src.storage._colcount
        if colcount is not None:
            colcount = colcount[start:start + length]
        storage = SparseStorage(row=row, col=col, value=_take(value, pos),
                                sparse_sizes=(src.sparse_size(0), length), colptr=colptr,
                                colcount=colcount, is_sorted=True, trust_data=True)
        return src.from_storage(storage)

    value = src.storage.value()
    if value is None:
        raise ValueError
    return src.set_value(value.narrow(dim - 1, start, length), layout='coo')


def __narrow_diag__(src: SparseTensor, start: Tuple[int, int], length: Tuple[int, int]) -> SparseTensor:
    """Inverse of ``cat_diag``: only for diagonally stacked matrices (reference narrow.py:83-130)."""
    rowptr, col, value = src.csr()
    lo, hi = torch.stack([rowptr[start[0]], rowptr[start[0] + length[0]]]).tolist()
    rowptr = rowptr[start[0]:start[0] + length[0] + 1] - lo
    row = src.storage._row
    if row is not None:
        row = row[lo:hi] - start[0]
    col = col[lo:hi] - start[1]
    if value is not None:
        value = value[lo:hi]
    st = src.storage
    rowcount = None if st._rowcount is None else st._rowcount[start[0]:start[0] + length[0]]
    colptr = None if st._colptr is None else st._colptr[start[1]:start[1] + length[1] + 1] - lo
    colcount = None if st._colcount is None else st._colcount[start[1]:start[1] + length[1]]
    csr2csc = None if st._csr2csc is None else st._csr2csc[lo:hi] - lo
    csc2csr = None if st._csc2csr is None else st._csc2csr[lo:hi] - lo
    storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                            sparse_sizes=(length[0], length[1]), rowcount=rowcount, colptr=colptr,
                            colcount=colcount, csr2csc=csr2csc, csc2csr=csc2csr, is_sorted=True,
                            trust_data=True)
    return src.from_storage(storage)


def select(src: SparseTensor, dim: int, idx: int) -> SparseTensor:
    return narrow(src, dim, start=idx, length=1)


def index_select(src: SparseTensor, dim: int, idx: Tensor) -> SparseTensor:
    dim = src.dim() + dim if dim < 0 else dim
    assert idx.dim() == 1

    if dim == 0:  # reference index_select.py:13-40
        _gpu(src)
        rowptr, col, value = src.csr()
        out_ptr, row, col, pos = torch.ops.tsamd.select_segments(rowptr, col, idx, True, True)
        storage = SparseStorage(row=row, rowptr=out_ptr, col=col, value=_take(value, pos),
                                sparse_sizes=(idx.size(0), src.sparse_size(1)), is_sorted=True,
                                trust_data=True)
        return src.from_storage(storage)

    if dim == 1:  # reference index_select.py:42-70: pick columns of the CSC view, re-sort to CSR
        _gpu(src)
        st = src.storage
        K = idx.size(0)
        if K > 0 and bool(((idx[1:] > idx[:-1]).all() & (idx[0] >= 0) & (idx[-1] < src.sparse_size(1)))):
            # strictly increasing ids (a sorted subset, e.g. nonzero(mask)): the selection keeps the
            # stored order, so it is a column mask -- one compaction, no CSC view, no re-sort
            mask = torch.zeros(src.sparse_size(1), dtype=torch.bool, device=idx.device)
            mask[idx] = True
            return masked_select(src, 1, mask)
        csr2csc = st.csr2csc()
        row_csc = st.row().index_select(0, csr2csc)
        colptr, col, row, pos = torch.ops.tsamd.select_segments(st.colptr(), row_csc, idx, True, True)
        row, col, csc2csr = torch.ops.tsamd.sort_coo(row, col, src.sparse_size(0), max(K, 1), True)
        value = st.value()
        if value is not None:
            value = value.index_select(0, csr2csc.index_select(0, pos.index_select(0, csc2csr)))
        storage = SparseStorage(row=row, col=col, value=value, sparse_sizes=(src.sparse_size(0), K),
                                colptr=colptr, csc2csr=csc2csr, is_sorted=True, trust_data=True)
        return src.from_storage(storage)

    value = src.storage.value()
    if value is None:
        raise ValueError
    return src.set_value(value.index_select(dim - 1, idx), layout='coo')


def index_select_nnz(src: SparseTensor, idx: Tensor, layout: Optional[str] = None) -> SparseTensor:
    assert idx.dim() == 1
    if get_layout(layout) == 'csc':
        idx = src.storage.csc2csr()[idx]
    row, col, value = src.coo()
    return SparseTensor(row=row.index_select(0, idx), col=col.index_select(0, idx),
                        value=_take(value, idx), sparse_sizes=src.sparse_sizes(), is_sorted=True)


def masked_select(src: SparseTensor, dim: int, mask: Tensor) -> SparseTensor:
    dim = src.dim() + dim if dim < 0 else dim
    assert mask.dim() == 1

    if dim in (0, 1):  # reference masked_select.py:15-63; kept rows / columns are renumbered by rank
        _gpu(src)
        assert mask.dtype == torch.bool and mask.numel() == src.sparse_size(dim)
        row, col, value = src.coo()
        pred = 'mask_row' if dim == 0 else 'mask_col'
        row, col, pos, kept = torch.ops.tsamd.filter_coo(pred, row, col, mask, 0, 0, True, 0, 0, True,
                                                         True)
        sizes = (kept, src.sparse_size(1)) if dim == 0 else (src.sparse_size(0), kept)
        storage = SparseStorage(row=row, col=col, value=_take(value, pos), sparse_sizes=sizes,
                                is_sorted=True, trust_data=True)
        return src.from_storage(storage)

    value = src.storage.value()
    if value is None:
        raise ValueError
    idx = mask.nonzero().flatten()
    return src.set_value(value.index_select(dim - 1, idx), layout='coo')


def masked_select_nnz(src: SparseTensor, mask: Tensor, layout: Optional[str] = None) -> SparseTensor:
    assert mask.dim() == 1
    _gpu(src)
    if get_layout(layout) == 'csc':
        mask = mask[src.storage.csc2csr()]
    row, col, value = src.coo()
    assert mask.dtype == torch.bool and mask.numel() == row.numel()
    row, col, pos, _ = torch.ops.tsamd.filter_coo('mask', row, col, mask, 0, 0, False, 0, 0, True, True)
    return SparseTensor(row=row, col=col, value=_take(value, pos), sparse_sizes=src.sparse_sizes(),
                        is_sorted=True)


def permute(src: SparseTensor, perm: Tensor) -> SparseTensor:
    """out[i, j] = src[perm[i], perm[j]] (reference permute.py:5-7: two index_selects).  When `perm`
    is a true permutation every entry (r, c) simply moves to (inv[r], inv[c]): two gathers and ONE
    radix sort instead of the CSC detour of index_select(1) (4 ms instead of 11 ms for 40 M entries)."""
    assert src.is_quadratic()
    n = src.sparse_size(0)
    if perm.dim() == 1 and perm.numel() == n and src.storage.col().is_cuda:
        inv = torch.full((n, ), -1, dtype=torch.long, device=perm.device)
        inv[perm] = torch.arange(n, device=perm.device)
        if not bool((inv < 0).any()):  # a permutation: nothing repeated, nothing left out
            row, col, value = src.coo()
            return SparseTensor(row=inv.index_select(0, row), col=inv.index_select(0, col), value=value,
                                sparse_sizes=(n, n), is_sorted=False, trust_data=True)
    return src.index_select(0, perm).index_select(1, perm)


def __getitem__(self: SparseTensor, index: Any) -> SparseTensor:
    """Same indexing grammar as the reference (tensor.py:624-671): ints, slices without step,
    bool / long tensors, lists, numpy arrays and one Ellipsis."""
    index = list(index) if isinstance(index, tuple) else [index]
    if len([i for i in index if not isinstance(i, (Tensor, np.ndarray)) and i is Ellipsis]) > 1:
        raise SyntaxError

    dim = 0
    out = self
    while len(index) > 0:
        item = index.pop(0)
        if isinstance(item, (list, tuple)):
            item = torch.tensor(item, device=self.device())
        if isinstance(item, np.ndarray):
            item = torch.from_numpy(item).to(self.device())

        if isinstance(item, int):
            out = out.select(dim, item)
            dim += 1
        elif isinstance(item, slice):
            if item.step is not None:
                raise ValueError('Step parameter not yet supported.')
            start = 0 if item.start is None else item.start
            start = self.size(dim) + start if start < 0 else start
            stop = self.size(dim) if item.stop is None else item.stop
            stop = self.size(dim) + stop if stop < 0 else stop
            out = out.narrow(dim, start, max(stop - start, 0))
            dim += 1
        elif torch.is_tensor(item):
            if item.dtype == torch.bool:
                out = out.masked_select(dim, item)
                dim += 1
            elif item.dtype == torch.long:
                out = out.index_select(dim, item)
                dim += 1
        elif item is Ellipsis:
            if self.dim() - len(index) < dim:
                raise SyntaxError
            dim = self.dim() - len(index)
        else:
            raise SyntaxError
    return out


SparseTensor.narrow = lambda self, dim, start, length: narrow(self, dim, start, length)
SparseTensor.__narrow_diag__ = lambda self, start, length: __narrow_diag__(self, start, length)
SparseTensor.select = lambda self, dim, idx: select(self, dim, idx)
SparseTensor.index_select = lambda self, dim, idx: index_select(self, dim, idx)
SparseTensor.index_select_nnz = lambda self, idx, layout=None: index_select_nnz(self, idx, layout)
SparseTensor.masked_select = lambda self, dim, mask: masked_select(self, dim, mask)
SparseTensor.masked_select_nnz = lambda self, mask, layout=None: masked_select_nnz(self, mask, layout)
SparseTensor.permute = lambda self, perm: permute(self, perm)
SparseTensor.__getitem__ = __getitem__

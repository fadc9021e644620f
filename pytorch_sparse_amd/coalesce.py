"""Functional coalesce on raw COO tensors (API of torch_sparse/coalesce.py:5-25).

Runs straight on the fused ops, without building a SparseStorage: one order probe, at most one
radix sort (decided on the device), one head-flag/scan/compaction pass, ONE host sync (the output size),
and a segmented reduction that reads the values through the sort permutation (the permuted value tensor
is never materialised).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .segment import segment_reduce

_OPS = ('add', 'sum', 'mean', 'min', 'max')
_OPCODE = {'add': 0, 'sum': 0, 'mean': 1, 'min': 2, 'max': 3}  # TSAMD_SUM .. TSAMD_MAX (include/tsamd.h)
_FUSE_REDUCE = True  # (A/B scripts set it to False: tsamd::sort_coalesce + tsamd::segment_reduce, the round-6 mid route)


def _rides(value: Optional[Tensor], nnz: int) -> bool:
    """Plain 4- / 8-byte numbers without a gradient travel with their entries through the sort (4-byte ones through
    every pass, 8-byte ones gathered by the last pass: csrc/sort.hip) instead of being read through the permutation."""
    return (value is not None and value.dim() == 1 and value.size(0) == nnz and value.is_cuda and
            value.element_size() in (4, 8) and not value.requires_grad and not value.is_complex())


def sorted_unique(row: Tensor, col: Tensor, m: int, n: int, value: Optional[Tensor] = None):
    """-> (row_u, col_u, perm or None, seg_ptr or None, nnz_u[, value in sorted order or None]): the distinct
    (row, col) pairs in row-major order plus what is needed to reduce the values of duplicates.  seg_ptr is None
    when the input had no duplicates (then perm alone reorders the values).  With `value` (see _rides) the sixth
    result holds the values in sorted order (perm then need not be applied to them)."""
    nnz = col.numel()
    if value is not None:
        if nnz <= 1 or not _rides(value, nnz):
            return sorted_unique(row, col, m, n) + (None, )
        row_s, col_s, perm, counts, value_s = torch.ops.tsamd.sort_coo_values(row, col, m, n, 1, None, value)
        row_u, col_u, seg_ptr, n_dev = torch.ops.tsamd.coalesce_index(row_s, col_s)
        descents, n_u = torch.cat([counts[:1], n_dev]).tolist()  # the one host sync
        perm_opt: Optional[Tensor] = perm if descents > 0 else None
        if descents == 0:
            row_s, col_s, value_s = row, col, value
        if n_u == nnz:
            return row_s, col_s, perm_opt, None, nnz, value_s
        return row_u[:n_u], col_u[:n_u], perm_opt, seg_ptr, n_u, value_s
    if nnz <= 1:
        return row, col, None, None, nnz
    # everything is enqueued without looking at the data -- order probe, a radix sort that returns at once when
    # the probe finds no descent (tsamd::sort_coo_auto), head flags + scan + compaction -- and ONE transfer brings
    # back what the output size depends on: (#descents of the input, #distinct pairs)
    row_s, col_s, perm, counts = torch.ops.tsamd.sort_coo_auto(row, col, m, n)
    row_u, col_u, seg_ptr, n_dev = torch.ops.tsamd.coalesce_index(row_s, col_s)
    descents, n_u = torch.cat([counts[:1], n_dev]).tolist()  # the one host sync
    perm_opt: Optional[Tensor] = perm if descents > 0 else None
    if descents == 0:
        row_s, col_s = row, col  # already in order: hand the caller's own tensors back (as the reference does)
    if n_u == nnz:
        return row_s, col_s, perm_opt, None, nnz
    return row_u[:n_u], col_u[:n_u], perm_opt, seg_ptr, n_u


def coalesce_rows_cols(row: Tensor, col: Tensor, value: Optional[Tensor], m: int, n: int,
                       op: str = 'add') -> Tuple[Tensor, Optional[Tensor]]:
    """`coalesce` on the two index rows given separately (the functional transpose hands them over swapped, without
    stacking them first)."""
    if op not in _OPS:
        raise ValueError(op)
    nnz = col.numel()
    if (_FUSE_REDUCE and nnz > 1 and col.is_cuda and
            (value is None or (_rides(value, nnz) and value.dtype in (torch.float32, torch.int32)))):
        # sort + duplicate compaction + reduction of the duplicates' values in one op (tsamd_sort_coalesce_reduce):
        # when the bucket sort takes the input, the distinct pairs AND their reduced values are written straight from
        # its last kernel (no run starts, no sorted copy of the values, no reduction pass); ONE transfer brings back
        # (#descents, #duplicates, #distinct, reduced-on-the-device flag)
        index_u, seg_ptr, counts, value_s, value_u = torch.ops.tsamd.sort_coalesce_reduce(row, col, m, n, value,
                                                                                        _OPCODE[op])
        descents, _, n_u, fused = counts.tolist()  # the one host sync
        if n_u == nnz:
            if descents == 0:  # in order, no duplicates: the caller's own data (as the reference hands it back)
                return torch.stack([row, col], dim=0), value
            return index_u, ((value_u if fused else value_s) if value is not None else None)
        if value is None:  # index only (the bucket route then wrote no run starts either)
            return index_u[:, :n_u].contiguous(), None
        if fused:  # (a prefix of the capacity-nnz buffer; its own storage once half of it would be idle)
            value = value_u[:n_u] if 2 * n_u >= nnz else value_u[:n_u].clone()
        else:  # the values came out of the sort in order: a streamed reduction, no gather
            value = segment_reduce(value_s, None, seg_ptr, n_u, op, balanced=nnz > 8 * max(n_u, 1))
        return index_u[:, :n_u].contiguous(), value
    if nnz > 1 and col.is_cuda and (value is None or _rides(value, nnz)):
        # sort + duplicate compaction in one op (tsamd_sort_coalesce): when the bucket sort takes the input the distinct
        # pairs are written straight from its last kernel; ONE transfer brings back (#descents, #duplicates, #distinct)
        index_u, seg_ptr, counts, value_s = torch.ops.tsamd.sort_coalesce(row, col, m, n, value)
        descents, _, n_u = counts.tolist()  # the one host sync
        if n_u == nnz:
            if descents == 0:  # in order, no duplicates: the caller's own data (as the reference hands it back)
                return torch.stack([row, col], dim=0), value
            return index_u, (value_s if value is not None else None)
        if value is not None:  # the values came out of the sort in order: a streamed reduction, no gather
            value = segment_reduce(value_s, None, seg_ptr, n_u, op, balanced=nnz > 8 * max(n_u, 1))
        return index_u[:, :n_u].contiguous(), value
    row, col, perm, seg_ptr, n_u, value_s = sorted_unique(row, col, m, n, value) if value is not None else (
        sorted_unique(row, col, m, n) + (None, ))
    if value_s is not None:  # the values came out of the sort in order: a streamed reduction, no gather
        value = value_s
        if seg_ptr is not None:
            value = segment_reduce(value_s, None, seg_ptr, n_u, op, balanced=nnz > 8 * max(n_u, 1))
    elif value is not None:
        if seg_ptr is not None:
            # differentiable, like segment_csr; long runs of duplicates go the entry-balanced way
            value = segment_reduce(value, perm, seg_ptr, n_u, op, balanced=nnz > 8 * max(n_u, 1))
        elif perm is not None:
            value = value.index_select(0, perm)
    return torch.stack([row, col], dim=0), value


def coalesce(index: Tensor, value: Optional[Tensor], m: int, n: int,
             op: str = 'add') -> Tuple[Tensor, Optional[Tensor]]:
    """Sort `index` ([2, nnz]) row-major and merge duplicate entries, reducing their values
    ([nnz, *], any supported dtype) with `op` in add | sum | mean | min | max."""
    return coalesce_rows_cols(index[0], index[1], value, m, n, op)

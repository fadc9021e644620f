"""Functional coalesce (reference: torch_sparse/coalesce.py:5-25)."""
import torch

from .storage import SparseStorage


def coalesce(index, value, m, n, op='add'):
    """Sort (index, value) row-major and merge duplicate entries with `op`
    (add | sum | mean | min | max).  Returns (index [2, nnz'], value)."""
    storage = SparseStorage(row=index[0], col=index[1], value=value, sparse_sizes=(m, n),
                            is_sorted=False)
    storage = storage.coalesce(reduce=op)
    return torch.stack([storage.row(), storage.col()], dim=0), storage.value()

"""Diagonal editing: ``remove_diag``, ``set_diag``, ``fill_diag``, ``get_diag`` (reference:
torch_sparse/diag.py).  ``fill_diag(adj, 1)`` is the self-loop step of GCN normalisation, i.e. the op
right before the ``sum(dim=1)`` / ``mul`` / SpMM chain.

``remove_diag`` is one ``tsamd::filter_coo`` compaction; ``set_diag`` drops the old diagonal entries and
merges the full k-th diagonal in one fused pass (``tsamd::set_diag_pattern``: flags + scan, one move
kernel with closed-form slots, one binary-search kernel for the diagonal) instead of the reference's
``remove_diag`` + ``non_diag_mask`` + four boolean-mask scatters; the values of old and new entries are
assembled with one differentiable gather.  ``torch.ops.torch_sparse.non_diag_mask`` itself is also
provided with the reference's schema.
"""
from typing import Optional

import torch
from torch import Tensor

from .storage import SparseStorage
from .tensor import SparseTensor


def remove_diag(src: SparseTensor, k: int = 0) -> SparseTensor:
    row, col, value = src.coo()
    if not col.is_cuda:
        raise RuntimeError('pytorch_sparse_amd has no CPU implementation: move the SparseTensor to the GPU')
    row, col, pos, _ = torch.ops.tsamd.filter_coo('off_diag', row, col, None, k, 0, False, 0, 0, True, True)
    if value is not None:
        value = value.index_select(0, pos)
    storage = SparseStorage(row=row, col=col, value=value, sparse_sizes=src.sparse_sizes(),
                            is_sorted=True, trust_data=True)
    return src.from_storage(storage)


def set_diag(src: SparseTensor, values: Optional[Tensor] = None, k: int = 0) -> SparseTensor:
    """Reference diag.py:37-79 (remove_diag, non_diag_mask, four boolean-mask scatters), fused: one
    flag + scan pass, one move kernel, one kernel for the diagonal, one gather of the values."""
    row, col, value = src.coo()
    if not col.is_cuda:
        raise RuntimeError('pytorch_sparse_amd has no CPU implementation: move the SparseTensor to the GPU')
    M, N = src.sparse_sizes()
    new_row, new_col, gather = torch.ops.tsamd.set_diag_pattern(row, col, M, N, k)
    num_diag = max(min(M + k, N) if k < 0 else min(M, N - k), 0)

    new_value: Optional[Tensor] = None
    if value is not None:
        shape = (num_diag, ) + tuple(value.size()[1:])
        if values is None:
            diag = value.new_ones(shape)
        else:
            diag = torch.as_tensor(values, dtype=value.dtype, device=value.device).expand(shape)
        new_value = torch.cat([value, diag], dim=0).index_select(0, gather)

    storage = SparseStorage(row=new_row, col=new_col, value=new_value, sparse_sizes=(M, N),
                            is_sorted=True, trust_data=True)
    return src.from_storage(storage)


def fill_diag(src: SparseTensor, fill_value: float, k: int = 0) -> SparseTensor:
    num_diag = min(src.sparse_size(0), src.sparse_size(1) - k)
    if k < 0:
        num_diag = min(src.sparse_size(0) + k, src.sparse_size(1))
    value = src.storage.value()
    if value is not None:
        sizes = [max(num_diag, 0)] + src.sizes()[2:]
        return set_diag(src, value.new_full(sizes, fill_value), k)
    return set_diag(src, None, k)


def get_diag(src: SparseTensor) -> Tensor:
    row, col, value = src.coo()
    if value is None:
        value = torch.ones(row.size(0), device=row.device)
    sizes = list(value.size())
    sizes[0] = min(src.size(0), src.size(1))
    # off-diagonal entries are parked in one extra slot that is dropped afterwards (no host sync)
    out = value.new_zeros([sizes[0] + 1] + sizes[1:])
    slot = torch.where(row == col, row, torch.full_like(row, sizes[0]))
    # differentiable w.r.t. the stored values, like the reference's `out[row[mask]] = value[mask]`
    # (torch_sparse/diag.py:98-110): index_copy into fresh zeros carries the gradient of its source
    out = out.index_copy(0, slot, value)
    return out[:sizes[0]]


SparseTensor.remove_diag = lambda self, k=0: remove_diag(self, k)
SparseTensor.set_diag = lambda self, values=None, k=0: set_diag(self, values, k)
SparseTensor.fill_diag = lambda self, fill_value, k=0: fill_diag(self, fill_value, k)
SparseTensor.get_diag = lambda self: get_diag(self)

"""Relabelled ("channel-camping free") layout, end to end.

``matmul`` on a Kronecker-like graph (R-MAT, most social / web graphs) fixes the HBM channel camping
of its hub columns by copying the dense operand to hashed row positions on EVERY call -- 15 % of a
north-star step.  A caller that multiplies with the same matrix many times (every
layer and epoch of a GNN) can keep its dense matrices in that layout for good::

    x_h = to_relabelled(x)                       # once: x_h[h(i)] = x[i],  h = relabel_index(N)
    y_h = matmul_relabelled(adj, x_h, 'sum')     # = to_relabelled(adj.matmul(x)), bit for bit;
    z_h = matmul_relabelled(adj, relu(y_h))      #   no probe, no copy; feeds the next layer as is
    z   = from_relabelled(z_h)                   # once

The hashed column ids are computed once per matrix and cached next to its storage.  Values are
bit-identical to the plain path (only addresses change); ``arg`` results hold the original entry
ids.  sum / mean are differentiable w.r.t. the dense operand and the sparse values (the gradients
live in the same layout); min / max are forward only here.  No reference counterpart: the reference
has no layout control at all (torch_sparse/matmul.py:9-91 is the plain path, ``matmul`` here).
"""
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from .tensor import SparseTensor

_COL_H = weakref.WeakKeyDictionary()  # SparseStorage -> hashed column ids (same order as col)


def relabel_index(n: int, device) -> Tensor:
    """h [n] int64: row i of a plain [n, K] matrix is row h[i] of its relabelled form."""
    like = torch.empty(0, dtype=torch.long, device=device)
    return torch.ops.tsamd.relabel_ids(None, n, like)


def to_relabelled(x: Tensor, dim: int = -2) -> Tensor:
    h = relabel_index(x.size(dim), x.device)
    return torch.empty_like(x).index_copy_(dim, h, x)


def from_relabelled(x_h: Tensor, dim: int = -2) -> Tensor:
    return x_h.index_select(dim, relabel_index(x_h.size(dim), x_h.device))


def _col_h(src: SparseTensor) -> Tensor:
    st = src.storage
    col_h = _COL_H.get(st)
    if col_h is None:
        col = st.col()
        col_h = torch.ops.tsamd.relabel_ids(col, st.sparse_size(1), col)
        _COL_H[st] = col_h
    return col_h


def matmul_relabelled(src: SparseTensor, other_h: Tensor, reduce: str = 'sum',
                      return_arg: bool = False):
    """``to_relabelled(src.matmul(from_relabelled(other_h), reduce))`` without leaving the layout."""
    if reduce == 'add':
        reduce = 'sum'
    if reduce not in ('sum', 'mean', 'min', 'max'):
        raise ValueError(reduce)
    st = src.storage
    rowptr, _, value = src.csr()
    if value is not None:
        value = value.to(other_h.dtype)
    row = rowcount = colptr = csr2csc = None
    if reduce in ('sum', 'mean'):
        if value is not None and value.requires_grad:
            row = st.row()
        if other_h.requires_grad:
            row, csr2csc, colptr = st.row(), st.csr2csc(), st.colptr()
        if reduce == 'mean' and row is not None:
            rowcount = st.rowcount()
    out, arg = torch.ops.tsamd.spmm_relabelled(row, rowptr, _col_h(src), value, rowcount, colptr, csr2csc,
                                               other_h, reduce)
    if return_arg:
        return out, (arg if reduce in ('min', 'max') else None)
    return out

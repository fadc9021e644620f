"""Row / column / global reductions of a SparseTensor (reference: torch_sparse/reduce.py).

SURVEY.md section 8f, rank 2: the step either side of SpMM in GCN-style normalisation
(``deg = adj.sum(dim=1)``).  Row reductions are a segmented reduction over ``rowptr``; column
reductions read the values through ``csr2csc`` and segment over ``colptr`` -- both through
``tsamd::segment_reduce`` (the reference uses torch_scatter's segment_csr / scatter).  Empty rows /
columns give 0.  Differentiable w.r.t. the sparse values like the reference (segment.py).
"""
from typing import Optional

import torch

from .segment import segment_reduce
from .tensor import SparseTensor


def _op(name: str) -> str:
    if name not in ('sum', 'add', 'mean', 'min', 'max'):
        raise ValueError(name)
    return name


def reduction(src: SparseTensor, dim: Optional[int] = None, reduce: str = 'sum') -> torch.Tensor:
    reduce = _op(reduce)
    value = src.storage.value()
    if dim is None:
        if value is None:
            n = src.nnz() if reduce in ('sum', 'add') else 1
            return torch.tensor(n, dtype=src.dtype(), device=src.device())
        return {'sum': value.sum, 'add': value.sum, 'mean': value.mean, 'min': value.min,
                'max': value.max}[reduce]()
    if dim < 0:
        dim = src.dim() + dim
    if dim > 1:
        if value is None:
            raise ValueError
        d = dim - 1
        return {'sum': lambda: value.sum(dim=d), 'add': lambda: value.sum(dim=d),
                'mean': lambda: value.mean(dim=d), 'min': lambda: value.min(dim=d)[0],
                'max': lambda: value.max(dim=d)[0]}[reduce]()
    st = src.storage
    if value is None:  # pattern only: counts, or ones
        if reduce in ('sum', 'add'):
            return (st.rowcount() if dim == 1 else st.colcount()).to(src.dtype())
        return torch.ones(src.size(dim == 0), dtype=src.dtype(), device=src.device())
    if dim == 1:
        return segment_reduce(value, None, st.rowptr(), src.size(0), reduce, balanced=True)
    return segment_reduce(value, st.csr2csc(), st.colptr(), src.size(1), reduce, balanced=True)


def sum(src: SparseTensor, dim: Optional[int] = None) -> torch.Tensor:
    return reduction(src, dim, 'sum')


def mean(src: SparseTensor, dim: Optional[int] = None) -> torch.Tensor:
    return reduction(src, dim, 'mean')


def min(src: SparseTensor, dim: Optional[int] = None) -> torch.Tensor:
    return reduction(src, dim, 'min')


def max(src: SparseTensor, dim: Optional[int] = None) -> torch.Tensor:
    return reduction(src, dim, 'max')


SparseTensor.sum = lambda self, dim=None: sum(self, dim)
SparseTensor.mean = lambda self, dim=None: mean(self, dim)
SparseTensor.min = lambda self, dim=None: min(self, dim)
SparseTensor.max = lambda self, dim=None: max(self, dim)

"""Pure-torch, TorchScript-compatible stand-in for rusty1s/pytorch_scatter, used ONLY by
tests/golden/make_golden.py to import the reference Python package in this container
(torch_scatter is a pip dependency of the reference and is not installed).
Implements the four entry points the reference calls: scatter, scatter_add, segment_csr,
gather_csr.  Empty segments are 0, integer 'mean' floors -- as torch_scatter does."""
from typing import Optional

import torch
from torch import Tensor

__version__ = '2.1.2'


def _broadcast(index: Tensor, src: Tensor, dim: int) -> Tensor:
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(0, dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def scatter(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
            dim_size: Optional[int] = None, reduce: str = 'sum') -> Tensor:
    assert out is None
    if dim < 0:
        dim = src.dim() + dim
    idx = _broadcast(index, src, dim)
    size = list(src.size())
    if dim_size is not None:
        size[dim] = dim_size
    elif idx.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(idx.max()) + 1
    res = torch.zeros(size, dtype=src.dtype, device=src.device)
    if reduce == 'sum' or reduce == 'add':
        return res.scatter_add_(dim, idx, src)
    if reduce == 'mean':
        res = res.scatter_add_(dim, idx, src)
        ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
        cnt = torch.zeros([size[dim]], dtype=src.dtype, device=src.device).scatter_add_(0, index, ones)
        cnt = _broadcast(cnt.clamp(min=1), res, dim)
        if res.is_floating_point():
            return res.div(cnt)
        return res.div(cnt, rounding_mode='floor')
    if reduce == 'min':
        return res.scatter_reduce(dim, idx, src, 'amin', include_self=False)
    if reduce == 'max':
        return res.scatter_reduce(dim, idx, src, 'amax', include_self=False)
    raise ValueError(reduce)


def scatter_add(src: Tensor, index: Tensor, dim: int = -1, out: Optional[Tensor] = None,
                dim_size: Optional[int] = None) -> Tensor:
    return scatter(src, index, dim, out, dim_size, 'sum')


def segment_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None,
                reduce: str = 'sum') -> Tensor:
    assert indptr.dim() == 1
    count = indptr[1:] - indptr[:-1]
    seg = torch.repeat_interleave(torch.arange(count.numel(), device=src.device), count)
    return scatter(src, seg, 0, None, count.numel(), reduce)


def gather_csr(src: Tensor, indptr: Tensor, out: Optional[Tensor] = None) -> Tensor:
    count = indptr[1:] - indptr[:-1]
    seg = torch.repeat_interleave(torch.arange(count.numel(), device=src.device), count)
    return src.index_select(0, seg)

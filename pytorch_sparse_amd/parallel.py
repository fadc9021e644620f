"""1-D row-sharded SpMM across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; its slicing primitive ``narrow(src, 0, start, length)``
(torch_sparse/narrow.py:15-42) defines what a row shard is: ``rowptr[start:start+length+1] -
rowptr[start]`` with the matching slices of ``col`` / ``value`` and *global* column ids.

Path per step (BASELINE.json north_star):
    X_full = all_gather(X_local)           RCCL, the only data-path collective
    out_local = A_local @ X_full           tsamd_spmm on the local row block
The output stays row-sharded, so there is no reduce step in the forward.  In the backward the
gradient of X is a partial sum on every rank and is reduce-scattered to its owners; the gradient
of the sparse values is local.

``spmm_fn`` is injectable so that the sharding / collective logic can be exercised on CPU with the
gloo backend (tests/test_parallel_cpu.py); the product default is the HIP operator.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def partition_rows(rowptr: Tensor, parts: int, balance: str = 'nnz') -> List[Tuple[int, int]]:
    """Contiguous row ranges [(start, end)] for `parts` ranks.

    balance='rows': equal row counts.  balance='nnz': split points at equal shares of the
    non-zeros (searchsorted on rowptr) -- what a power-law matrix needs."""
    M = rowptr.numel() - 1
    if balance == 'rows':
        cuts = [(M * p) // parts for p in range(parts + 1)]
    elif balance == 'nnz':
        E = int(rowptr[-1])
        targets = torch.tensor([(E * p) // parts for p in range(1, parts)], dtype=rowptr.dtype,
                               device=rowptr.device)
        inner = torch.searchsorted(rowptr, targets, right=False).clamp_(0, M).tolist() if parts > 1 else []
        cuts = [0] + inner + [M]
        for i in range(1, len(cuts)):  # monotone even with empty stretches
            cuts[i] = max(cuts[i], cuts[i - 1])
    else:
        raise ValueError(balance)
    return [(cuts[p], cuts[p + 1]) for p in range(parts)]


def narrow_rows(rowptr: Tensor, col: Tensor, value: Optional[Tensor], start: int, end: int):
    """Row block [start, end) as a local CSR with global column ids (views, no copies of col/value)."""
    e0, e1 = int(rowptr[start]), int(rowptr[end])
    return rowptr[start:end + 1] - e0, col[e0:e1], None if value is None else value[e0:e1]


def _default_spmm(rowptr, col, value, x, reduce):
    from .matmul import matmul
    from .tensor import SparseTensor
    A = SparseTensor(rowptr=rowptr, col=col, value=value, sparse_sizes=(rowptr.numel() - 1, x.size(-2)),
                     is_sorted=True, trust_data=True)
    return matmul(A, x, reduce)


class _GatherRows(torch.autograd.Function):
    """all_gather of row blocks of X; backward = reduce-scatter of the gradient to the owners."""

    @staticmethod
    def forward(ctx, x_local: Tensor, sizes: Sequence[int], group):
        ctx.sizes, ctx.group = list(sizes), group
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        ctx.rank = rank
        F = x_local.shape[1:]
        if len(set(sizes)) == 1:  # equal shards: one collective straight into the result
            out = x_local.new_empty((sum(sizes), ) + tuple(F))
            dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
            return out
        mx = max(sizes)
        pad = x_local.new_zeros((mx, ) + tuple(F))
        pad[:x_local.size(0)] = x_local
        buf = x_local.new_empty((world * mx, ) + tuple(F))
        dist.all_gather_into_tensor(buf, pad, group=group)
        return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)], 0)

    @staticmethod
    def backward(ctx, grad_full: Tensor):
        sizes, group, rank = ctx.sizes, ctx.group, ctx.rank
        grad_full = grad_full.contiguous()
        start = sum(sizes[:rank])
        if len(set(sizes)) == 1 and dist.get_backend(group) != 'gloo':
            out = grad_full.new_empty((sizes[rank], ) + tuple(grad_full.shape[1:]))
            dist.reduce_scatter_tensor(out, grad_full, group=group)
            return out, None, None
        dist.all_reduce(grad_full, group=group)  # gloo has no reduce_scatter
        return grad_full[start:start + sizes[rank]].clone(), None, None


class RowShardedSpMM(object):
    """The local row block of A plus the bookkeeping to multiply it with a row-sharded X.

    rowptr/col/value: local CSR (global column ids) -- e.g. from ``narrow_rows``.
    x_sizes: number of X rows owned by each rank (sum = number of columns of A).
    """

    def __init__(self, rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                 group=None, spmm_fn: Optional[Callable] = None):
        self.rowptr, self.col, self.value = rowptr, col, value
        self.x_sizes = list(x_sizes)
        self.group = group
        self.spmm_fn = spmm_fn or _default_spmm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(self.x_sizes) == self.world

    def gather(self, x_local: Tensor) -> Tensor:
        if self.world == 1:
            return x_local
        return _GatherRows.apply(x_local, self.x_sizes, self.group)

    def __call__(self, x_local: Tensor, reduce: str = 'sum') -> Tensor:
        """out_local [rows of this rank, F] = A_local @ all_gather(x_local)."""
        return self.spmm_fn(self.rowptr, self.col, self.value, self.gather(x_local), reduce)


def shard_matrix(rowptr: Tensor, col: Tensor, value: Optional[Tensor], n_cols: int, group=None,
                 balance: str = 'nnz', spmm_fn: Optional[Callable] = None) -> Tuple[RowShardedSpMM, Tuple[int, int]]:
    """Convenience for a replicated global CSR: every rank cuts out its own row block.  X is
    sharded by the same row ranges when the matrix is square (GNN layers chain that way),
    otherwise in equal blocks of columns."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ranges = partition_rows(rowptr, world, balance)
    M = rowptr.numel() - 1
    if M == n_cols:
        x_sizes = [e - s for s, e in ranges]
    else:
        x_sizes = [(n_cols * (p + 1)) // world - (n_cols * p) // world for p in range(world)]
    s, e = ranges[rank]
    rp, c, v = narrow_rows(rowptr, col, value, s, e)
    return RowShardedSpMM(rp, c, v, x_sizes, group, spmm_fn), (s, e)

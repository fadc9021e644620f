"""One-hop neighbour sampling (reference: torch_sparse/sample.py; csrc/cpu/sample_cpu.cpp is CPU-only
there -- "No CUDA version supported", csrc/sample.cpp:21-23).  SURVEY.md 8f rank 4: the producer of the
``adj_t`` every GraphSAGE-style mini-batch multiplies with.

``sample_adj`` runs entirely on the GPU (``torch.ops.torch_sparse.sample_adj``: count + scan, one draw
per lane, dense first-occurrence relabel, per-row radix sort; csrc/sample.hip).  Seeded through torch's
CPU generator (``torch.manual_seed``).  Which neighbours are drawn necessarily differs from the
reference's generator; everything that is not random is identical: row ``i`` of the result belongs to
``subset[i]``, ``n_id`` starts with ``subset`` and continues with the new nodes in first-occurrence
order, rows are sorted by the new column id, ``e_id`` are positions into the source ``col`` / ``value``;
``num_neighbors < 0`` (take every neighbour) is bit-identical.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .tensor import SparseTensor


def sample(src: SparseTensor, num_neighbors: int, subset: Optional[Tensor] = None) -> Tensor:
    """``num_neighbors`` uniform draws WITH replacement per row -> [rows, num_neighbors] column ids
    (reference sample.py:7-25, the same five tensor ops; rows without neighbours read out of range
    there, here they give -1)."""
    rowptr, col, _ = src.csr()
    rowcount = src.storage.rowcount()
    if subset is not None:
        rowcount = rowcount[subset]
        rowptr = rowptr[subset]
    else:
        rowptr = rowptr[:-1]
    rand = torch.rand((rowcount.size(0), num_neighbors), device=col.device)
    rand.mul_(rowcount.to(rand.dtype).view(-1, 1))
    rand = rand.to(torch.long)
    rand = torch.minimum(rand, (rowcount.view(-1, 1) - 1).clamp_(min=0))
    rand.add_(rowptr.view(-1, 1))
    out = col[rand.clamp_(max=max(col.numel() - 1, 0))] if col.numel() > 0 else rand.new_full(rand.shape, -1)
    return out.masked_fill_((rowcount == 0).view(-1, 1), -1)


def sample_adj(src: SparseTensor, subset: Tensor, num_neighbors: int,
               replace: bool = False) -> Tuple[SparseTensor, Tensor]:
    rowptr, col, value = src.csr()
    rowptr, col, n_id, e_id = torch.ops.torch_sparse.sample_adj(rowptr, col, subset, num_neighbors,
                                                               replace)
    if value is not None:
        value = value.index_select(0, e_id)
    out = SparseTensor(rowptr=rowptr, row=None, col=col, value=value,
                       sparse_sizes=(subset.size(0), n_id.size(0)), is_sorted=True, trust_data=True)
    return out, n_id


SparseTensor.sample = sample
SparseTensor.sample_adj = sample_adj

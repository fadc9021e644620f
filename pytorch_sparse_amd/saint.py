"""GraphSAINT sub-graph extraction (reference: torch_sparse/saint.py, csrc/cpu/saint_cpu.cpp -- CPU
only there).  The sub-graph induced by ``node_idx``, nodes renumbered by their position in
``node_idx``; rows follow ``node_idx``, every row keeps its stored column order, ``edge_index`` are
the positions of the kept entries in the source.  Runs as select + filter + remap on the GPU
(``torch.ops.torch_sparse.saint_subgraph``), bit-identical to the reference."""
from typing import Tuple

import torch
from torch import Tensor

from .tensor import SparseTensor


def saint_subgraph(src: SparseTensor, node_idx: Tensor) -> Tuple[SparseTensor, Tensor]:
    row, col, value = src.coo()
    rowptr = src.storage.rowptr()
    row, col, edge_index = torch.ops.torch_sparse.saint_subgraph(node_idx, rowptr, row, col)
    if value is not None:
        value = value.index_select(0, edge_index)
    out = SparseTensor(row=row, rowptr=None, col=col, value=value,
                       sparse_sizes=(node_idx.size(0), node_idx.size(0)), is_sorted=True, trust_data=True)
    return out, edge_index


SparseTensor.saint_subgraph = saint_subgraph

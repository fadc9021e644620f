"""METIS partitioning entry point (reference: torch_sparse/metis.py).  The reference only partitions
when it was compiled WITH_METIS and otherwise raises "Not compiled with METIS support"
(csrc/cpu/metis_cpu.cpp:60); no METIS library exists for this build, so `partition` keeps the trivial
single-part case of the reference and raises that same error for everything else."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .tensor import SparseTensor


def partition(src: SparseTensor, num_parts: int, recursive: bool = False, weighted: bool = False,
              node_weight: Optional[Tensor] = None, balance_edge: bool = False
              ) -> Tuple[SparseTensor, Tensor, Tensor]:
    assert num_parts >= 1
    if num_parts == 1:
        partptr = torch.tensor([0, src.size(0)], device=src.device())
        perm = torch.arange(src.size(0), device=src.device())
        return src, partptr, perm
    raise RuntimeError('Not compiled with METIS support')


SparseTensor.partition = partition

"""Reverse Cuthill-McKee reordering (reference: torch_sparse/bandwidth.py).  As in the reference the
ordering itself is scipy's (``scipy.sparse.csgraph.reverse_cuthill_mckee`` on the host); the
symmetrisation before it and the permutation after it run on the GPU."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .select import permute
from .tensor import SparseTensor


def reverse_cuthill_mckee(src: SparseTensor, is_symmetric: Optional[bool] = None) -> Tuple[SparseTensor, Tensor]:
    import scipy.sparse as sp
    if is_symmetric is None:
        is_symmetric = src.is_symmetric()
    if not is_symmetric:
        src = src.to_symmetric()
    sp_src = src.to_scipy(layout='csr')
    perm = sp.csgraph.reverse_cuthill_mckee(sp_src, symmetric_mode=True).copy()
    perm = torch.from_numpy(perm).to(torch.long).to(src.device())
    return permute(src, perm), perm


SparseTensor.reverse_cuthill_mckee = reverse_cuthill_mckee

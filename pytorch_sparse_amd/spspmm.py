"""Legacy functional SpSpMM (reference: torch_sparse/spspmm.py:6-33)."""
import torch

from .matmul import matmul
from .tensor import SparseTensor


def spspmm(indexA, valueA, indexB, valueB, m, k, n, coalesced=False):
    """C = A * B for COO inputs (index [2, nnz], value).  Inputs must be coalesced; pass
    ``coalesced=True`` to have them sorted first.  Returns (indexC [2, nnzC], valueC)."""
    A = SparseTensor(row=indexA[0], col=indexA[1], value=valueA, sparse_sizes=(m, k),
                     is_sorted=not coalesced)
    B = SparseTensor(row=indexB[0], col=indexB[1], value=valueB, sparse_sizes=(k, n),
                     is_sorted=not coalesced)
    C = matmul(A, B)
    row, col, value = C.coo()
    return torch.stack([row, col], dim=0), value

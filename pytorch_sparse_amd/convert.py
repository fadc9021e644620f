"""Legacy (index, value) <-> torch sparse / scipy converters (reference: torch_sparse/convert.py,
torch_sparse/eye.py, torch_sparse/spadd.py).  Host-side glue of the functional API; ``spadd`` runs
through the package's GPU ``coalesce``."""
import numpy as np
import torch

from .coalesce import coalesce


def to_torch_sparse(index, value, m, n):
    return torch.sparse_coo_tensor(index.detach(), value, (m, n))


def from_torch_sparse(A):
    return A.indices().detach(), A.values()


def to_scipy(index, value, m, n):
    import scipy.sparse
    assert not index.is_cuda and not value.is_cuda
    (row, col), data = index.detach(), value.detach()
    return scipy.sparse.coo_matrix((data, (row, col)), (m, n))


def from_scipy(A):
    A = A.tocoo()
    row = torch.from_numpy(A.row.astype(np.int64))
    col = torch.from_numpy(A.col.astype(np.int64))
    return torch.stack([row, col], dim=0), torch.from_numpy(A.data)


def eye(m, dtype=None, device=None):
    """(index, value) of the m x m identity."""
    row = torch.arange(m, dtype=torch.long, device=device)
    return torch.stack([row, row], dim=0), torch.ones(m, dtype=dtype, device=device)


def spadd(indexA, valueA, indexB, valueB, m, n):
    """Sum of two sparse matrices given as (index, value) pairs."""
    index = torch.cat([indexA, indexB], dim=-1)
    value = torch.cat([valueA, valueB], dim=0)
    return coalesce(index=index, value=value, m=m, n=n, op='add')

"""t() on SparseTensor and the functional transpose (reference: torch_sparse/transpose.py)."""
import torch

from .storage import SparseStorage
from .tensor import SparseTensor


def t(src: SparseTensor) -> SparseTensor:
    """Transpose by permuting with csr2csc (one cached radix sort) and swapping the CSR/CSC caches,
    so ``A.t().t()`` costs nothing more (reference transpose.py:7-31)."""
    st = src.storage
    perm = st.csr2csc()
    row, col, value = src.coo()
    M, N = st.sparse_sizes()
    out = SparseStorage(row=col[perm], rowptr=st._colptr, col=row[perm],
                        value=None if value is None else value[perm], sparse_sizes=(N, M),
                        rowcount=st._colcount, colptr=st._rowptr, colcount=st._rowcount,
                        csr2csc=st._csc2csr, csc2csr=perm, is_sorted=True, trust_data=True)
    return src.from_storage(out)


SparseTensor.t = lambda self: t(self)


def transpose(index, value, m, n, coalesced=True):
    """Functional transpose of a COO matrix given as (index [2, nnz], value); with
    ``coalesced=True`` the result is sorted row-major and duplicates are summed
    (reference transpose.py:39-62)."""
    row, col = index[1], index[0]
    if coalesced:
        storage = SparseStorage(row=row, col=col, value=value, sparse_sizes=(n, m), is_sorted=False)
        storage = storage.coalesce()
        row, col, value = storage.row(), storage.col(), storage.value()
    return torch.stack([row, col], dim=0), value

"""t() on SparseTensor and the functional transpose (API of torch_sparse/transpose.py)."""
import torch

from .coalesce import coalesce_rows_cols
from .storage import SparseStorage
from .tensor import SparseTensor


def t(src: SparseTensor) -> SparseTensor:
    """Transpose by permuting with csr2csc (one cached radix sort) and handing the CSC-side caches
    over as the CSR-side caches of the result, so ``A.t().t()`` costs nothing more
    (reference transpose.py:7-31)."""
    st = src.storage
    # sorted (col, row) -- and the values in that order -- straight from the sort when csr2csc is new
    col_t, row_t, perm, value_t = st.csc_index_value()
    M, N = st.sparse_sizes()
    out = SparseStorage(row=col_t, rowptr=st._colptr, col=row_t,
                        value=value_t, sparse_sizes=(N, M),
                        rowcount=st._colcount, colptr=st._rowptr, colcount=st._rowcount,
                        csr2csc=st._csc2csr, csc2csr=perm, is_sorted=True, trust_data=True)
    return src.from_storage(out)


SparseTensor.t = lambda self: t(self)


def transpose(index, value, m, n, coalesced=True):
    """(index, value) of the n x m transpose.  With ``coalesced=True`` (default) the result is
    sorted row-major with duplicates summed -- i.e. a coalesce of the swapped index
    (reference transpose.py:39-62); otherwise only the two index rows are swapped."""
    if not coalesced:
        return torch.stack([index[1], index[0]], dim=0), value
    return coalesce_rows_cols(index[1], index[0], value, n, m, op='add')  # (the swapped rows, never stacked)

"""SparseTensor: the object API over SparseStorage (reference: torch_sparse/tensor.py).

Covers construction, format views (coo/csr/csc), caches, dtype/device plumbing, dense and
torch.sparse conversions, and -- attached by the sibling modules, as the reference does --
``matmul/spmm/spspmm/@``, ``t()``, ``coalesce()``, reductions, element-wise ``mul/add``, diagonal
edits, ``narrow/select/index_select/masked_select/permute/[]``, ``sample/sample_adj``,
``random_walk``, ``saint_subgraph`` and ``reverse_cuthill_mckee``.  Not provided: ``hgt_sample``,
``ego_k_hop_sample_adj`` and the METIS partitioner; the heterogeneous / temporal neighbour
samplers are operators (``torch.ops.torch_sparse.hetero_neighbor_sample`` / ``hetero_temporal_neighbor_sample``).

Like the reference's (tensor.py:12) the class is a TorchScript class: ``torch.jit.script`` functions and
modules can take, build and return it (``matmul(adj, x, reduce)`` inside a scripted ``nn.Module``,
``torch.jit.script(spspmm)`` -- reference test/test_matmul.py:79).  The class body compiles; what cannot
(``to(*args)``, scipy conversions, ``__repr__``, pinning) is attached below it.
"""
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from .storage import SparseStorage, get_layout


def storage_spmm(st: SparseStorage, other: Tensor, reduce: str,
                 want_arg: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """``A @ other`` on a storage -> (out, arg_out for min / max).  Collects the cached arrays the
    reference front-end hands to its ops (torch_sparse/matmul.py:12-28, 38-56, 60-77): the CSC-side
    caches are only filled when a gradient w.r.t. `other` will be asked for."""
    rowptr, col, value = st.rowptr(), st.col(), st.value()
    if value is not None:
        value = value.to(other.dtype)
    if reduce == 'min' or reduce == 'max':
        # The CSC arrays (a radix sort on first use) are only built when the pull backward will really run: a
        # gradient w.r.t. `other` is being recorded and the rows are wide enough for the pull (winner records +
        # masked sum that skips segments without winners [+ masked SDDMM for grad_value]) to beat the scatter kernel
        # (packed atomics, grad_value fused), or deterministic algorithms are asked for.  Same-box table, 2^20-row
        # R-MAT graph, ms pull / scatter (profiles/r05_minmax_bw_route_rule.md):
        #   value-less   bf16 K = 32: 1.52 / 0.77   64: 1.28 / 1.42   128: 1.37 / 2.47   256: 2.27 / 4.19
        #                fp32 K = 32: 1.57 / 0.93   64: 1.32 / 1.70   128: 1.60 / 2.83   256: 2.81 / 5.30
        #   + grad_value bf16 K = 32: 1.71 / 1.47   64: 1.52 / 1.57   128: 1.81 / 2.5-2.9  256: 2.96 / 4.76
        #                fp32 K = 32: 1.77 / 1.66   64: 1.69 / 1.85   128: 2.22 / 2.97   256: 3.87 / 5.90
        # (f16 K = 64 with grad_value: 1.66 / 1.34 -- two-byte rows take the pull from 128 features on.)
        pull = other.requires_grad and torch.is_grad_enabled()
        if pull and not torch.ops.tsamd.deterministic():
            k = other.size(-1)
            narrow = other.dtype == torch.bfloat16 or other.dtype == torch.float16
            if value is not None and value.requires_grad and narrow:
                pull = k >= 128
            else:
                pull = k >= 64
        if pull:
            # training: hand the CSC arrays over (cached in the storage, as for sum) so that grad_mat is
            # pulled column by column instead of scattered with atomics (tsamd_spmm_minmax_bw_csc)
            # want_arg False (matmul / spmm return `out` alone): the winners stay inside the autograd node, as
            # 32-bit ids -- half the bytes in the forward's store and in the backward's read (include/tsamd.h)
            out, arg = torch.ops.tsamd.spmm_minmax(rowptr, col, value, st.colptr(), st.csr2csc(), st.row(),
                                                   other, reduce == 'max', not want_arg)
            return out, arg
        if reduce == 'min':
            out, arg = torch.ops.torch_sparse.spmm_min(rowptr, col, value, other)
            return out, arg
        out, arg = torch.ops.torch_sparse.spmm_max(rowptr, col, value, other)
        return out, arg
    row, csr2csc, colptr, rowcount = st._row, st._csr2csc, st._colptr, st._rowcount
    if value is not None and value.requires_grad:
        row = st.row()
    if other.requires_grad:
        row, csr2csc, colptr = st.row(), st.csr2csc(), st.colptr()
        if reduce == 'mean':
            rowcount = st.rowcount()
    none: Optional[Tensor] = None
    if reduce == 'sum' or reduce == 'add':
        return torch.ops.tsamd.spmm_sum_owned(row, rowptr, col, value, colptr, csr2csc, other), none
    if reduce == 'mean':
        return torch.ops.tsamd.spmm_mean_owned(row, rowptr, col, value, rowcount, colptr, csr2csc,
                                                other), none
    raise ValueError


@torch.jit.script
class SparseTensor(object):
    storage: SparseStorage

    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None,
                 col: Optional[Tensor] = None, value: Optional[Tensor] = None,
                 sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
                 is_sorted: bool = False, trust_data: bool = False):
        self.storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                                     sparse_sizes=sparse_sizes, rowcount=None, colptr=None,
                                     colcount=None, csr2csc=None, csc2csr=None, is_sorted=is_sorted,
                                     trust_data=trust_data)

    # ---- constructors ------------------------------------------------------------------------
    @classmethod
    def from_storage(self, storage: SparseStorage):
        # TorchScript classes have no __new__: build an instance on the storage's OWN tensors (no new
        # allocations, no launches -- is_sorted / trust_data skip every check that reads the device), then swap
        # the storage in so that its caches and its pending-sort state come along (reference tensor.py:36-47)
        out = SparseTensor(row=storage._row, rowptr=storage._rowptr, col=storage._col, value=storage._value,
                           sparse_sizes=storage._sparse_sizes, is_sorted=True, trust_data=True)
        out.storage = storage
        return out

    @classmethod
    def from_edge_index(self, edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                        sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
                        is_sorted: bool = False, trust_data: bool = False):
        return SparseTensor(row=edge_index[0], rowptr=None, col=edge_index[1], value=edge_attr,
                            sparse_sizes=sparse_sizes, is_sorted=is_sorted, trust_data=trust_data)

    @classmethod
    def from_dense(self, mat: Tensor, has_value: bool = True):
        if mat.dim() > 2:
            index = mat.abs().sum([i for i in range(2, mat.dim())]).nonzero()
        else:
            index = mat.nonzero()
        index = index.t()
        row, col = index[0], index[1]
        value: Optional[Tensor] = None
        if has_value:
            value = mat[row, col]
        return SparseTensor(row=row, rowptr=None, col=col, value=value,
                            sparse_sizes=(mat.size(0), mat.size(1)), is_sorted=True, trust_data=True)

    @classmethod
    def from_torch_sparse_coo_tensor(self, mat: Tensor, has_value: bool = True):
        mat = mat.coalesce()
        index = mat._indices()
        value: Optional[Tensor] = None
        if has_value:
            value = mat.values()
        return SparseTensor(row=index[0], rowptr=None, col=index[1], value=value,
                            sparse_sizes=(mat.size(0), mat.size(1)), is_sorted=True, trust_data=True)

    @classmethod
    def from_torch_sparse_csr_tensor(self, mat: Tensor, has_value: bool = True):
        value: Optional[Tensor] = None
        if has_value:
            value = mat.values()
        return SparseTensor(row=None, rowptr=mat.crow_indices(), col=mat.col_indices(), value=value,
                            sparse_sizes=(mat.size(0), mat.size(1)), is_sorted=True, trust_data=True)

    @classmethod
    def eye(self, M: int, N: Optional[int] = None, has_value: bool = True,
            dtype: Optional[int] = None, device: Optional[torch.device] = None,
            fill_cache: bool = False):
        n: int = M
        if N is not None:
            n = N
        k = min(M, n)
        idx = torch.arange(k, device=device)
        rowptr = torch.cat([torch.arange(k + 1, device=device),
                            torch.full((M - k, ), k, dtype=torch.long, device=device)])
        value: Optional[Tensor] = None
        if has_value:
            value = torch.ones(k, dtype=dtype, device=device)
        out = SparseTensor(row=idx, rowptr=rowptr, col=idx, value=value, sparse_sizes=(M, n),
                           is_sorted=True, trust_data=True)
        if fill_cache:
            out.storage.fill_cache_()
        return out

    # ---- views -------------------------------------------------------------------------------
    def copy(self):
        return self.from_storage(self.storage)

    def clone(self):
        return self.from_storage(self.storage.clone())

    def coo(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        return self.storage.row(), self.storage.col(), self.storage.value()

    def csr(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        return self.storage.rowptr(), self.storage.col(), self.storage.value()

    def csc(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        perm = self.storage.csr2csc()
        value = self.storage.value()
        if value is not None:
            value = value.index_select(0, perm)
        return self.storage.colptr(), self.storage.row().index_select(0, perm), value

    def has_value(self) -> bool:
        return self.storage.has_value()

    def set_value_(self, value: Optional[Tensor], layout: Optional[str] = None):
        self.storage.set_value_(value, layout)
        return self

    def set_value(self, value: Optional[Tensor], layout: Optional[str] = None):
        return self.from_storage(self.storage.set_value(value, layout))

    def fill_value_(self, fill_value: float, dtype: Optional[torch.dtype] = None):
        value = torch.full((self.nnz(), ), fill_value, dtype=dtype, device=self.device())
        return self.set_value_(value, layout='coo')

    def fill_value(self, fill_value: float, dtype: Optional[torch.dtype] = None):
        value = torch.full((self.nnz(), ), fill_value, dtype=dtype, device=self.device())
        return self.set_value(value, layout='coo')

    # ---- products (dense operand; the Python-level ``matmul`` / ``@`` attached by matmul.py also take a
    #      SparseTensor) -- real methods, so that scripted code can call ``adj.matmul(x, reduce)`` ------
    def spmm(self, other: Tensor, reduce: str = 'sum') -> Tensor:
        return storage_spmm(self.storage, other, reduce, False)[0]

    def matmul(self, other: Tensor, reduce: str = 'sum') -> Tensor:
        return storage_spmm(self.storage, other, reduce, False)[0]

    # ---- sizes -------------------------------------------------------------------------------
    def sparse_sizes(self) -> Tuple[int, int]:
        return self.storage.sparse_sizes()

    def sparse_size(self, dim: int) -> int:
        return self.storage.sparse_sizes()[dim]

    def sparse_resize(self, sparse_sizes: Tuple[int, int]):
        return self.from_storage(self.storage.sparse_resize(sparse_sizes))

    def sparse_reshape(self, num_rows: int, num_cols: int):
        return self.from_storage(self.storage.sparse_reshape(num_rows, num_cols))

    def sizes(self) -> List[int]:
        sizes = list(self.sparse_sizes())
        value = self.storage.value()
        if value is not None:
            sizes += list(value.size())[1:]
        return sizes

    def size(self, dim: int) -> int:
        return self.sizes()[dim]

    def dim(self) -> int:
        return len(self.sizes())

    def nnz(self) -> int:
        return self.storage._col.numel()

    def numel(self) -> int:
        value = self.storage.value()
        return value.numel() if value is not None else self.nnz()

    def density(self) -> float:
        M, N = self.sparse_sizes()
        return 0.0 if M == 0 or N == 0 else self.nnz() / (M * N)

    def sparsity(self) -> float:
        return 1 - self.density()

    def avg_row_length(self) -> float:
        return self.nnz() / self.sparse_size(0)

    def avg_col_length(self) -> float:
        return self.nnz() / self.sparse_size(1)

    def bandwidth(self) -> int:
        row, col, _ = self.coo()
        return int((row - col).abs_().max())

    def avg_bandwidth(self) -> float:
        row, col, _ = self.coo()
        return float((row - col).abs_().to(torch.float).mean())

    def bandwidth_proportion(self, bandwidth: int) -> float:
        row, col, _ = self.coo()
        return int(((row - col).abs_() <= bandwidth).sum()) / self.nnz()

    def is_quadratic(self) -> bool:
        return self.sparse_size(0) == self.sparse_size(1)

    def is_symmetric(self) -> bool:
        if not self.is_quadratic():
            return False
        rowptr, col, value1 = self.csr()
        colptr, row, value2 = self.csc()
        if (rowptr != colptr).any() or (col != row).any():
            return False
        if value1 is None or value2 is None:
            return True
        return bool((value1 == value2).all())

    def to_symmetric(self, reduce: str = 'sum'):
        """A + A^T with duplicates merged (reference tensor.py:404-437), via the fused sort/coalesce."""
        N = max(self.sparse_size(0), self.sparse_size(1))
        row, col, value = self.coo()
        storage = SparseStorage(row=torch.cat([row, col]), col=torch.cat([col, row]),
                                value=None if value is None else torch.cat([value, value]),
                                sparse_sizes=(N, N), is_sorted=False, trust_data=True)
        return self.from_storage(storage.coalesce(reduce=reduce))

    # ---- coalescing / caches -------------------------------------------------------------------
    def is_coalesced(self) -> bool:
        return self.storage.is_coalesced()

    def coalesce(self, reduce: str = 'sum'):
        return self.from_storage(self.storage.coalesce(reduce))

    def fill_cache_(self):
        self.storage.fill_cache_()
        return self

    def clear_cache_(self):
        self.storage.clear_cache_()
        return self

    def __eq__(self, other) -> bool:
        if not isinstance(other, self.__class__):
            return False
        if self.sizes() != other.sizes():
            return False
        rowptr1, col1, value1 = self.csr()
        rowptr2, col2, value2 = other.csr()
        if (value1 is None) != (value2 is None):
            return False
        if rowptr1.numel() != rowptr2.numel() or col1.numel() != col2.numel():
            return False
        if not (torch.equal(rowptr1, rowptr2) and torch.equal(col1, col2)):
            return False
        return value1 is None or torch.equal(value1, value2)

    # ---- autograd / device / dtype -------------------------------------------------------------
    def detach_(self):
        value = self.storage.value()
        if value is not None:
            value.detach_()
        return self

    def detach(self):
        value = self.storage.value()
        return self.set_value(value.detach() if value is not None else None, layout='coo')

    def requires_grad(self) -> bool:
        value = self.storage.value()
        return value.requires_grad if value is not None else False

    def requires_grad_(self, requires_grad: bool = True, dtype: Optional[torch.dtype] = None):
        if requires_grad and not self.has_value():
            self.fill_value_(1., dtype)
        value = self.storage.value()
        if value is not None:
            value.requires_grad_(requires_grad)
        return self

    def device(self):
        return self.storage._col.device

    def is_cuda(self) -> bool:
        return self.storage._col.is_cuda

    def dtype(self):
        value = self.storage.value()
        return value.dtype if value is not None else torch.float

    def is_floating_point(self) -> bool:
        value = self.storage.value()
        return torch.is_floating_point(value) if value is not None else True

    def type(self, dtype: torch.dtype, non_blocking: bool = False):
        value = self.storage.value()
        if value is None or dtype == value.dtype:
            return self
        return self.from_storage(self.storage.type(dtype, non_blocking))

    def type_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.type(tensor.dtype, non_blocking)

    def to_device(self, device: torch.device, non_blocking: bool = False):
        if device == self.device():
            return self
        return self.from_storage(self.storage.to_device(device, non_blocking))

    def device_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.to_device(tensor.device, non_blocking)

    def bfloat16(self):
        return self.type(torch.bfloat16, False)

    def half(self):
        return self.type(torch.half, False)

    def float(self):
        return self.type(torch.float, False)

    def double(self):
        return self.type(torch.double, False)

    def int(self):
        return self.type(torch.int, False)

    def long(self):
        return self.type(torch.long, False)

    def bool(self):
        return self.type(torch.bool, False)

    def byte(self):
        return self.type(torch.uint8, False)

    def char(self):
        return self.type(torch.int8, False)

    def short(self):
        return self.type(torch.short, False)

    # ---- conversions ---------------------------------------------------------------------------
    def to_dense(self, dtype: Optional[torch.dtype] = None) -> Tensor:
        row, col, value = self.coo()
        if value is not None:
            mat = torch.zeros(self.sizes(), dtype=value.dtype, device=self.device())
            mat[row, col] = value
        else:
            mat = torch.zeros(self.sizes(), dtype=dtype, device=self.device())
            mat[row, col] = torch.ones(self.nnz(), dtype=mat.dtype, device=mat.device)
        return mat

    def to_torch_sparse_coo_tensor(self, dtype: Optional[torch.dtype] = None) -> Tensor:
        row, col, value = self.coo()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_coo_tensor(torch.stack([row, col], dim=0), value, self.sizes())

    def to_torch_sparse_csr_tensor(self, dtype: Optional[torch.dtype] = None) -> Tensor:
        rowptr, col, value = self.csr()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_csr_tensor(rowptr, col, value, self.sizes())

    def to_torch_sparse_csc_tensor(self, dtype: Optional[torch.dtype] = None) -> Tensor:
        colptr, row, value = self.csc()
        if value is None:
            value = torch.ones(self.nnz(), dtype=dtype, device=self.device())
        return torch.sparse_csc_tensor(colptr, row, value, self.sizes())


# ---- Python-only methods (not visible to TorchScript; the reference attaches its own the same way) --
def _to(self, *args, **kwargs):
    device, dtype, non_blocking = torch._C._nn._parse_to(*args, **kwargs)[:3]
    out = self
    if dtype is not None:
        out = out.type(dtype, non_blocking)
    if device is not None:
        out = out.to_device(device, non_blocking)
    return out


def _cuda(self, device: Optional[Union[int, str]] = None, non_blocking: bool = False):
    return self.to_device(torch.device('cuda' if device is None else device), non_blocking)


def _share_memory_(self):
    self.storage.share_memory_()
    return self


def _from_scipy(mat, has_value: bool = True):
    colptr = None
    if mat.format == 'csc':
        colptr = torch.from_numpy(mat.indptr).to(torch.long)
    mat = mat.tocsr()
    rowptr = torch.from_numpy(mat.indptr).to(torch.long)
    mat = mat.tocoo()
    row = torch.from_numpy(mat.row).to(torch.long)
    col = torch.from_numpy(mat.col).to(torch.long)
    value = torch.from_numpy(mat.data) if has_value else None
    storage = SparseStorage(row=row, rowptr=rowptr, col=col, value=value,
                            sparse_sizes=tuple(mat.shape), colptr=colptr, is_sorted=True,
                            trust_data=True)
    return SparseTensor.from_storage(storage)


def _to_scipy(self, layout: Optional[str] = None, dtype: Optional[torch.dtype] = None):
    import scipy.sparse
    assert self.dim() == 2
    layout = get_layout(layout)
    ones = lambda: torch.ones(self.nnz(), dtype=dtype)  # noqa: E731
    if layout == 'coo':
        row, col, value = self.coo()
        value = ones() if value is None else value.detach().cpu()
        return scipy.sparse.coo_matrix((value, (row.cpu(), col.cpu())), self.sizes())
    if layout == 'csr':
        rowptr, col, value = self.csr()
        value = ones() if value is None else value.detach().cpu()
        return scipy.sparse.csr_matrix((value, col.cpu(), rowptr.cpu()), self.sizes())
    colptr, row, value = self.csc()
    value = ones() if value is None else value.detach().cpu()
    return scipy.sparse.csc_matrix((value, row.cpu(), colptr.cpu()), self.sizes())


def _repr(self) -> str:
    row, col, value = self.coo()
    lines = ['row=%s' % row, 'col=%s' % col]
    if value is not None:
        lines.append('val=%s' % value)
    lines.append('size=%s, nnz=%d, density=%.2f%%' % (tuple(self.sizes()), self.nnz(),
                                                      100 * self.density()))
    return '%s(%s)' % (self.__class__.__name__, ',\n             '.join(lines))


SparseTensor.to = _to
SparseTensor.cpu = lambda self: self.to_device(torch.device('cpu'))
SparseTensor.cuda = _cuda
SparseTensor.pin_memory = lambda self: SparseTensor.from_storage(self.storage.pin_memory())
SparseTensor.is_pinned = lambda self: self.storage.is_pinned()
SparseTensor.share_memory_ = _share_memory_
SparseTensor.is_shared = lambda self: self.storage.is_shared()
SparseTensor.from_scipy = staticmethod(_from_scipy)
SparseTensor.to_scipy = _to_scipy
SparseTensor.__repr__ = _repr
# __getitem__, narrow, index_select, ... are attached by select.py / cat.py / diag.py / mul.py /
# reduce.py / sample.py, like the reference attaches its method modules

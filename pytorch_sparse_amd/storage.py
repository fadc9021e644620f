"""SparseStorage: row-major sorted COO plus lazily cached CSR/CSC views.

Same role and method names as ``torch_sparse/storage.py`` of the reference; the work is done by
fused HIP ops instead of Python/ATen/torch_scatter compositions:

  * sort-on-construct (reference storage.py:149-162)  -> ``tsamd::coo_check`` (order probe + range check /
    size inference in one pass, the single host sync) + one ``tsamd::sort_coo`` radix sort that emits sorted
    row/col and the permutation; with ``trust_data=True`` and both sizes given NO host sync at all: the sort
    is decided on the device (``tsamd::sort_coo_auto``);
  * ``csr2csc`` (storage.py:407-416)                   -> ``tsamd::sort_coo`` on (col, row);
  * ``coalesce`` (storage.py:436-466)                  -> ``tsamd::coalesce_index`` +
    ``tsamd::segment_reduce`` (values are read through the permutation, duplicates reduced in
    stable input order);
  * ``rowptr`` / ``row`` / ``colptr``                  -> ``torch_sparse::ind2ptr`` / ``ptr2ind``.

Like the reference's (storage.py:21) this is a TorchScript class: scripted functions and modules can
take, build and return it.  Everything in the class body compiles; the few Python-only conveniences
(pinning, shared memory) are attached below it, as the reference attaches its own.
"""
import warnings
from typing import List, Optional, Tuple

import torch
from torch import Tensor

layouts: List[str] = ['coo', 'csr', 'csc']
_CACHE_KEYS: List[str] = ['rowcount', 'colptr', 'colcount', 'csr2csc', 'csc2csr']

# what _remade() does to every tensor of a storage
_OP_SAME, _OP_CLONE, _OP_DEVICE = 0, 1, 2


def get_layout(layout: Optional[str] = None) -> str:
    if layout is None:
        layout = 'coo'
        warnings.warn('`layout` argument unset, using default layout "coo". This may lead to '
                      'unexpected behaviour.')
    assert layout == 'coo' or layout == 'csr' or layout == 'csc'
    return layout


def _checked(t: Optional[Tensor], like: Tensor, numel: int) -> Optional[Tensor]:
    """An optional int64 index array of `numel` (< 0: any) entries on `like`'s device, contiguous."""
    if t is None:
        return None
    assert t.dtype == torch.long, 'index tensors must be int64'
    assert t.device == like.device, 'index tensor on a different device'
    assert t.dim() == 1, 'index tensors must be 1-D'
    if numel >= 0:
        assert t.numel() == numel, 'index tensor of unexpected length'
    return t.contiguous()


def _xf(t: Optional[Tensor], op: int, device: torch.device, non_blocking: bool) -> Optional[Tensor]:
    if t is None:
        return None
    if op == 1:
        return t.clone()
    if op == 2:
        return t.to(device, non_blocking=non_blocking)
    return t


@torch.jit.ignore
def _reduce_duplicates(value: Tensor, seg_ptr: Tensor, n: int, reduce: str, balanced: bool) -> Tensor:
    # differentiable w.r.t. value (a Python autograd.Function): runs outside the compiled graph
    from .segment import segment_reduce
    return segment_reduce(value, None, seg_ptr, n, reduce, balanced=balanced)


@torch.jit.script
class SparseStorage(object):
    _row: Optional[Tensor]
    _rowptr: Optional[Tensor]
    _col: Tensor
    _value: Optional[Tensor]
    _sparse_sizes: Tuple[int, int]
    _rowcount: Optional[Tensor]
    _colptr: Optional[Tensor]
    _colcount: Optional[Tensor]
    _csr2csc: Optional[Tensor]
    _csc2csr: Optional[Tensor]
    _pending_sort: bool

    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None,
                 col: Optional[Tensor] = None, value: Optional[Tensor] = None,
                 sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
                 rowcount: Optional[Tensor] = None, colptr: Optional[Tensor] = None,
                 colcount: Optional[Tensor] = None, csr2csc: Optional[Tensor] = None,
                 csc2csr: Optional[Tensor] = None, is_sorted: bool = False,
                 trust_data: bool = False):
        assert row is not None or rowptr is not None
        assert col is not None
        assert col.dtype == torch.long, 'col must be int64'
        assert col.dim() == 1, 'col must be 1-D'
        col = col.contiguous()
        nnz = col.numel()
        if value is not None:  # before anything (the sort below) hands `value` to a kernel under col's device guard
            assert value.device == col.device
            assert value.size(0) == nnz
        if row is not None:
            assert row.device == col.device

        # sizes: given, or inferred.  On the GPU everything the constructor may have to read back -- the order
        # probe of sort-on-construct and the two maxima of the range check / size inference -- comes from ONE
        # fused pass and ONE transfer (tsamd::coo_check); with trust_data=True and both sizes given nothing
        # is read back at all (the sort is then decided on the device, tsamd::sort_coo_auto).
        M: int = 0
        given_m: Optional[int] = None
        given_n: Optional[int] = None
        if sparse_sizes is not None:
            given_m, given_n = sparse_sizes
        need_row_max = rowptr is None and nnz > 0 and (given_m is None or not trust_data)
        need_col_max = nnz > 0 and (given_n is None or not trust_data)
        max_row: int = -1
        max_col: int = -1
        descents: int = -1  # unknown
        # Both sizes given, unsorted COO, no caller caches to preserve: the check, the device-decided sort and the
        # gather of the values are ENQUEUED back to back and the check's four words are read afterwards -- the one
        # read-back no longer sits between the probe and the sort (an out-of-range id is still reported, after a
        # sort whose result is then thrown away; ind2ptr clamps the ids it fills with to [0, M], csrc/convert.hip)
        presorted_row: Optional[Tensor] = None
        presorted_col: Optional[Tensor] = None
        presorted_value: Optional[Tensor] = None
        presorted_rowptr: Optional[Tensor] = None
        did_sort = False
        if (col.is_cuda and row is not None and rowptr is None and given_m is not None and given_n is not None and
                not trust_data and not is_sorted and nnz > 1 and csr2csc is None and csc2csr is None):
            # order probe + range check ride in the sort's first pass over (row, col); the values ride along in its
            # last pass when they are plain 4- / 8-byte numbers
            presorted_row, presorted_col, perm0, counts_dev, vsorted = torch.ops.tsamd.sort_coo_values(
                row.contiguous(), col, given_m, given_n, 3, None, value)
            if value is not None:
                presorted_value = vsorted
            # (rowptr too: CSR is what every product reads, and a launch behind the read-back would wait for it)
            presorted_rowptr = torch.ops.torch_sparse.ind2ptr(presorted_row, given_m)
            counts0: List[int] = counts_dev.tolist()  # the one host sync, with everything already in flight
            descents, max_row, max_col = counts0[0], counts0[2], counts0[3]
            did_sort = True
        elif col.is_cuda and row is not None and (need_row_max or need_col_max):
            counts: List[int] = torch.ops.tsamd.coo_check(row.contiguous(), col).tolist()  # the one host sync
            descents, max_row, max_col = counts[0], counts[2], counts[3]
        else:
            if need_row_max and row is not None:
                max_row = int(row.max())
            if need_col_max:
                max_col = int(col.max())
        if given_m is not None:
            M = given_m
            if rowptr is not None:
                assert rowptr.numel() - 1 == M
            elif need_row_max:
                assert max_row < M and (max_row >= 0 or not col.is_cuda)  # (the device check takes unsigned maxima: a negative id reads as -1)
        elif rowptr is not None:
            M = rowptr.numel() - 1
        elif row is not None and row.numel() > 0:
            M = max_row + 1
        N: int = 0
        if given_n is not None:
            N = given_n
            if need_col_max:
                assert max_col < N and (max_col >= 0 or not col.is_cuda)
        elif nnz > 0:
            N = max_col + 1
        self._sparse_sizes = (M, N)

        self._row = _checked(row, col, nnz)
        self._rowptr = _checked(rowptr, col, M + 1)
        self._col = col
        if value is not None:
            assert value.device == col.device
            assert value.size(0) == nnz
            value = value.contiguous()
        self._value = value
        self._rowcount = _checked(rowcount, col, M)
        self._colptr = _checked(colptr, col, N + 1)
        self._colcount = _checked(colcount, col, N)
        self._csr2csc = _checked(csr2csc, col, nnz)
        self._csc2csr = _checked(csc2csr, col, nnz)

        # Unsorted COO handed over on the CPU: there is no CPU sort in this package, so the sort is
        # deferred until the storage is moved to the GPU (`.cuda()` / `.to(device)`); until then every
        # accessor refuses to hand out the (still unsorted) arrays.
        self._pending_sort = (not is_sorted) and nnz > 1 and not col.is_cuda
        if did_sort:
            # (sorted or not, the outputs of the device-decided sort are what the storage holds)
            self._row = presorted_row
            self._rowptr = presorted_rowptr
            self._col = torch.jit._unwrap_optional(presorted_col)
            if value is not None:
                self._value = presorted_value
        elif (not is_sorted) and nnz > 1 and col.is_cuda:
            r = self.row()
            keep_caches = rowptr is not None or csr2csc is not None or csc2csr is not None
            if descents < 0 and keep_caches:
                # caller-supplied views of the pattern survive when the input turns out to be sorted (as in the
                # reference, storage.py:149-162): that needs the verdict on the host
                descents = int(torch.ops.tsamd.coo_order(r, col, N)[0])
            if descents < 0:
                # nothing was read back: sort decided on the device (a sorted input costs the probe, a few
                # kernels that return at once and one copy) -- no host sync in this constructor
                rs, cs, perm, _, vs = torch.ops.tsamd.sort_coo_values(r, col, M, N, 1, None, value)
                self._row = rs
                self._col = cs
                if value is not None:
                    self._value = vs
            elif descents > 0:
                rs, cs, perm, _, vs = torch.ops.tsamd.sort_coo_values(r, col, M, N, 0, None, value)
                self._row = rs
                self._col = cs
                self._rowptr = None
                if value is not None:
                    self._value = vs
                self._csr2csc = None
                self._csc2csr = None

    # ---- construction helpers --------------------------------------------------------------
    @classmethod
    def empty(self):
        z = torch.tensor([], dtype=torch.long)
        return SparseStorage(row=z, rowptr=None, col=z, value=None, sparse_sizes=(0, 0), rowcount=None,
                             colptr=None, colcount=None, csr2csc=None, csc2csr=None, is_sorted=True,
                             trust_data=True)

    def _remade(self, op: int, device: torch.device, non_blocking: bool, value: Optional[Tensor],
                keep_value: bool):
        """A new storage from this one's tensors, each one kept / cloned / moved (`op`); the values are
        this storage's (keep_value) or `value` (not transformed)."""
        self._ready_or_moving(op)
        v = value
        if keep_value:
            v = _xf(self._value, op, device, non_blocking)
        return SparseStorage(row=_xf(self._row, op, device, non_blocking),
                             rowptr=_xf(self._rowptr, op, device, non_blocking),
                             col=self._col if op == 0 else (self._col.clone() if op == 1 else
                                                            self._col.to(device, non_blocking=non_blocking)),
                             value=v, sparse_sizes=self._sparse_sizes,
                             rowcount=_xf(self._rowcount, op, device, non_blocking),
                             colptr=_xf(self._colptr, op, device, non_blocking),
                             colcount=_xf(self._colcount, op, device, non_blocking),
                             csr2csc=_xf(self._csr2csc, op, device, non_blocking),
                             csc2csr=_xf(self._csc2csr, op, device, non_blocking),
                             is_sorted=not self._pending_sort, trust_data=True)

    def _ready_or_moving(self, op: int):
        if op != 2:  # a pending CPU storage may only be moved (the move sorts it on the GPU)
            self._ready()

    # ---- COO / CSR views -------------------------------------------------------------------
    def _ready(self):
        if self._pending_sort:
            raise RuntimeError('this SparseStorage was built from unsorted COO on the CPU; '
                               'pytorch_sparse_amd has no CPU implementation -- move it to the GPU first '
                               '(.cuda() / .to(device) sorts it there) or pass is_sorted=True')

    def has_row(self) -> bool:
        return self._row is not None

    def row(self) -> Tensor:
        self._ready()
        row = self._row
        if row is not None:
            return row
        rowptr = self._rowptr
        if rowptr is None:
            raise ValueError
        row = torch.ops.torch_sparse.ptr2ind(rowptr, self._col.numel())
        self._row = row
        return row

    def has_rowptr(self) -> bool:
        return self._rowptr is not None

    def rowptr(self) -> Tensor:
        self._ready()
        rowptr = self._rowptr
        if rowptr is not None:
            return rowptr
        row = self._row
        if row is None:
            raise ValueError
        rowptr = torch.ops.torch_sparse.ind2ptr(row, self._sparse_sizes[0])
        self._rowptr = rowptr
        return rowptr

    def col(self) -> Tensor:
        self._ready()
        return self._col

    def has_value(self) -> bool:
        return self._value is not None

    def value(self) -> Optional[Tensor]:
        self._ready()
        return self._value

    def _layout_value(self, value: Optional[Tensor], layout: Optional[str]) -> Optional[Tensor]:
        if value is not None:
            if get_layout(layout) == 'csc':
                value = value[self.csc2csr()]
            value = value.contiguous()
            assert value.device == self._col.device
            assert value.size(0) == self._col.numel()
        return value

    def set_value_(self, value: Optional[Tensor], layout: Optional[str] = None):
        self._value = self._layout_value(value, layout)
        return self

    def set_value(self, value: Optional[Tensor], layout: Optional[str] = None):
        return self._remade(0, self._col.device, False, self._layout_value(value, layout), False)

    # ---- sizes -----------------------------------------------------------------------------
    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sparse_sizes

    def sparse_size(self, dim: int) -> int:
        return self._sparse_sizes[dim]

    def sparse_resize(self, sparse_sizes: Tuple[int, int]):
        self._ready()
        nnz = self._col.numel()
        rowptr, rowcount = self._rowptr, self._rowcount
        colptr, colcount = self._colptr, self._colcount
        grow_m = sparse_sizes[0] - self._sparse_sizes[0]
        if grow_m > 0:
            if rowptr is not None:
                rowptr = torch.cat([rowptr, rowptr.new_full((grow_m, ), nnz)])
            if rowcount is not None:
                rowcount = torch.cat([rowcount, rowcount.new_zeros(grow_m)])
        elif grow_m < 0:
            if rowptr is not None:
                rowptr = rowptr[:grow_m]
            if rowcount is not None:
                rowcount = rowcount[:grow_m]
        grow_n = sparse_sizes[1] - self._sparse_sizes[1]
        if grow_n > 0:
            if colptr is not None:
                colptr = torch.cat([colptr, colptr.new_full((grow_n, ), nnz)])
            if colcount is not None:
                colcount = torch.cat([colcount, colcount.new_zeros(grow_n)])
        elif grow_n < 0:
            if colptr is not None:
                colptr = colptr[:grow_n]
            if colcount is not None:
                colcount = colcount[:grow_n]
        return SparseStorage(row=self._row, rowptr=rowptr, col=self._col, value=self._value,
                             sparse_sizes=(sparse_sizes[0], sparse_sizes[1]), rowcount=rowcount,
                             colptr=colptr, colcount=colcount, csr2csc=self._csr2csc,
                             csc2csr=self._csc2csr, is_sorted=True, trust_data=True)

    def sparse_reshape(self, num_rows: int, num_cols: int):
        assert num_rows > 0 or num_rows == -1
        assert num_cols > 0 or num_cols == -1
        assert num_rows > 0 or num_cols > 0
        total = self._sparse_sizes[0] * self._sparse_sizes[1]
        if num_rows == -1:
            num_rows = total // num_cols
        if num_cols == -1:
            num_cols = total // num_rows
        assert num_rows * num_cols == total
        idx = self._sparse_sizes[1] * self.row() + self._col
        row = torch.div(idx, num_cols, rounding_mode='floor')
        col = idx - row * num_cols
        return SparseStorage(row=row, rowptr=None, col=col, value=self._value,
                             sparse_sizes=(num_rows, num_cols), rowcount=None, colptr=None,
                             colcount=None, csr2csc=None, csc2csr=None, is_sorted=True, trust_data=True)

    # ---- cached CSC-side views -------------------------------------------------------------
    def has_rowcount(self) -> bool:
        return self._rowcount is not None

    def rowcount(self) -> Tensor:
        rowcount = self._rowcount
        if rowcount is not None:
            return rowcount
        rowptr = self.rowptr()
        rowcount = rowptr[1:] - rowptr[:-1]
        self._rowcount = rowcount
        return rowcount

    def has_csr2csc(self) -> bool:
        return self._csr2csc is not None

    def csr2csc(self) -> Tensor:
        """Permutation that orders the entries column-major (stable radix sort of col*M+row)."""
        perm = self._csr2csc
        if perm is not None:
            return perm
        out = torch.ops.tsamd.sort_coo(self._col, self.row(), self._sparse_sizes[1], self._sparse_sizes[0],
                                       False)
        perm = out[2]
        self._csr2csc = perm
        return perm

    def csc_index(self) -> Tuple[Tensor, Tensor, Tensor]:
        """(col, row, csr2csc) with the entries in column-major order.  When the permutation is not cached
        yet, the radix sort that produces it also emits the sorted (col, row) pairs -- no gathers through the
        permutation afterwards (and colptr comes from the sorted columns for free); what `t()` needs."""
        perm = self._csr2csc
        if perm is not None:
            return self._col.index_select(0, perm), self.row().index_select(0, perm), perm
        N = self._sparse_sizes[1]
        cs, rs, perm = torch.ops.tsamd.sort_coo(self._col, self.row(), N, self._sparse_sizes[0], True)
        self._csr2csc = perm
        if self._colptr is None and self._colcount is None:
            self._colptr = torch.ops.torch_sparse.ind2ptr(cs, N)
        return cs, rs, perm

    def csc_index_value(self) -> Tuple[Tensor, Tensor, Tensor, Optional[Tensor]]:
        """``csc_index()`` plus the values in column-major order: when the permutation is new the values ride along
        in the sort that produces it (no gather through the permutation afterwards); what `t()` needs."""
        value = self._value
        if self._csr2csc is not None or value is None:
            cs, rs, perm = self.csc_index()
            return cs, rs, perm, None if value is None else value.index_select(0, perm)
        N = self._sparse_sizes[1]
        cs, rs, perm, _, vs = torch.ops.tsamd.sort_coo_values(self._col, self.row(), N, self._sparse_sizes[0], 0, None,
                                                              value)
        self._csr2csc = perm
        if self._colptr is None and self._colcount is None:
            self._colptr = torch.ops.torch_sparse.ind2ptr(cs, N)
        return cs, rs, perm, vs

    def has_csc2csr(self) -> bool:
        return self._csc2csr is not None

    def csc2csr(self) -> Tensor:
        inv = self._csc2csr
        if inv is not None:
            return inv
        perm = self.csr2csc()
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=perm.device)
        self._csc2csr = inv
        return inv

    def has_colptr(self) -> bool:
        return self._colptr is not None

    def colptr(self) -> Tensor:
        ptr = self._colptr
        if ptr is not None:
            return ptr
        N = self._sparse_sizes[1]
        colcount = self._colcount
        if colcount is not None:
            ptr = self._col.new_zeros(N + 1)
            torch.cumsum(colcount, dim=0, out=ptr[1:])
        else:
            perm0 = self._csr2csc
            if perm0 is not None:
                ptr = torch.ops.torch_sparse.ind2ptr(self._col.index_select(0, perm0), N)
            else:
                # a new csr2csc comes with the sorted columns: no gather through the permutation
                cs, rs, perm = self.csc_index()
                ptr2 = self._colptr
                if ptr2 is not None:
                    return ptr2
                ptr = torch.ops.torch_sparse.ind2ptr(cs, N)
        self._colptr = ptr
        return ptr

    def has_colcount(self) -> bool:
        return self._colcount is not None

    def colcount(self) -> Tensor:
        colcount = self._colcount
        if colcount is not None:
            return colcount
        ptr = self.colptr()
        colcount = ptr[1:] - ptr[:-1]
        self._colcount = colcount
        return colcount

    # ---- coalescing ------------------------------------------------------------------------
    def is_coalesced(self) -> bool:
        self._ready()
        if self._col.numel() <= 1:
            return True
        counts = torch.ops.tsamd.coo_order(self.row(), self._col, self._sparse_sizes[1])
        return int(counts[0]) == 0 and int(counts[1]) == 0

    def coalesce(self, reduce: str = 'add'):
        """Merge duplicate (row, col) entries; values of duplicates are reduced in storage order."""
        self._ready()
        nnz = self._col.numel()
        if nnz <= 1:
            return self
        row = self.row()
        dup = int(torch.ops.tsamd.coo_order(row, self._col, self._sparse_sizes[1])[1])
        if dup == 0:  # already coalesced
            return self
        row_u, col_u, seg_ptr, n_dev = torch.ops.tsamd.coalesce_index(row, self._col)
        n = nnz - dup
        value = self._value
        if value is not None:
            # heavy duplication (few distinct pairs, long runs): take the entry-balanced path
            value = _reduce_duplicates(value, seg_ptr, n, reduce, nnz > 8 * max(n, 1))
        return SparseStorage(row=row_u[:n].clone(), rowptr=None, col=col_u[:n].clone(), value=value,
                             sparse_sizes=self._sparse_sizes, rowcount=None, colptr=None, colcount=None,
                             csr2csc=None, csc2csr=None, is_sorted=True, trust_data=True)

    # ---- cache management ------------------------------------------------------------------
    def fill_cache_(self):
        self.row()
        self.rowptr()
        self.rowcount()
        self.csr2csc()
        self.csc2csr()
        self.colptr()
        self.colcount()
        return self

    def clear_cache_(self):
        self._rowcount = None
        self._colptr = None
        self._colcount = None
        self._csr2csc = None
        self._csc2csr = None
        return self

    def cached_keys(self) -> List[str]:
        keys: List[str] = []
        if self.has_rowcount():
            keys.append('rowcount')
        if self.has_colptr():
            keys.append('colptr')
        if self.has_colcount():
            keys.append('colcount')
        if self.has_csr2csc():
            keys.append('csr2csc')
        if self.has_csc2csr():
            keys.append('csc2csr')
        return keys

    def num_cached_keys(self) -> int:
        return len(self.cached_keys())

    # ---- copies / device / dtype -----------------------------------------------------------
    def copy(self):
        return self._remade(0, self._col.device, False, None, True)

    def clone(self):
        return self._remade(1, self._col.device, False, None, True)

    def type(self, dtype: torch.dtype, non_blocking: bool = False):
        value = self._value
        if value is None or dtype == value.dtype:
            return self
        return self.set_value(value.to(dtype=dtype, non_blocking=non_blocking), layout='coo')

    def type_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.type(tensor.dtype, non_blocking)

    def to_device(self, device: torch.device, non_blocking: bool = False):
        if device == self._col.device:
            return self
        return self._remade(2, device, non_blocking, None, True)

    def device_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.to_device(tensor.device, non_blocking)

    def cuda(self):
        if self._col.is_cuda:
            return self
        return self._remade(2, torch.device('cuda'), False, None, True)

    def cpu(self):
        if not self._col.is_cuda:
            return self
        return self._remade(2, torch.device('cpu'), False, None, True)

    def is_cuda(self) -> bool:
        return self._col.is_cuda


# ---- Python-only conveniences (not visible to TorchScript, like the reference's) -----------------
def _tensors(self) -> List[Tensor]:
    ts = [self._row, self._rowptr, self._col, self._value, self._rowcount, self._colptr, self._colcount,
          self._csr2csc, self._csc2csr]
    return [t for t in ts if t is not None]


def _pin_memory(self):
    def pin(t):
        return None if t is None else t.pin_memory()
    return SparseStorage(row=pin(self._row), rowptr=pin(self._rowptr), col=pin(self._col),
                         value=pin(self._value), sparse_sizes=self._sparse_sizes,
                         rowcount=pin(self._rowcount), colptr=pin(self._colptr),
                         colcount=pin(self._colcount), csr2csc=pin(self._csr2csc),
                         csc2csr=pin(self._csc2csr), is_sorted=not self._pending_sort, trust_data=True)


def _share_memory_(self):
    for t in _tensors(self):
        t.share_memory_()
    return self


SparseStorage.pin_memory = _pin_memory
SparseStorage.is_pinned = lambda self: all(t.is_pinned() for t in _tensors(self))
SparseStorage.share_memory_ = _share_memory_
SparseStorage.is_shared = lambda self: all(t.is_shared() for t in _tensors(self))

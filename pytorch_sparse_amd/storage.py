"""SparseStorage: row-major sorted COO plus lazily cached CSR/CSC views.

Same role and method names as ``torch_sparse/storage.py`` of the reference; the work is done by
fused HIP ops instead of Python/ATen/torch_scatter compositions:

  * sort-on-construct (reference storage.py:149-162)  -> one ``tsamd::coo_order`` probe (the single
    host sync) + one ``tsamd::sort_coo`` radix sort that emits sorted row/col and the permutation;
  * ``csr2csc`` (storage.py:407-416)                   -> ``tsamd::sort_coo`` on (col, row);
  * ``coalesce`` (storage.py:436-466)                  -> ``tsamd::coalesce_index`` +
    ``tsamd::segment_reduce`` (values are read through the permutation, duplicates reduced in
    stable input order);
  * ``rowptr`` / ``row`` / ``colptr``                  -> ``torch_sparse::ind2ptr`` / ``ptr2ind``.

The class is a plain Python object (not a TorchScript class).
"""
import warnings
from typing import List, Optional, Tuple

import torch
from torch import Tensor

layouts = ['coo', 'csr', 'csc']
_CACHE_KEYS = ('rowcount', 'colptr', 'colcount', 'csr2csc', 'csc2csr')


def get_layout(layout: Optional[str] = None) -> str:
    if layout is None:
        layout = 'coo'
        warnings.warn('`layout` argument unset, using default layout "coo". This may lead to '
                      'unexpected behaviour.')
    assert layout in layouts
    return layout


def _check_index(t: Optional[Tensor], name: str, like: Tensor, numel: Optional[int] = None):
    if t is None:
        return None
    assert t.dtype == torch.long, '%s must be int64' % name
    assert t.device == like.device, '%s is on a different device' % name
    assert t.dim() == 1, '%s must be 1-D' % name
    if numel is not None:
        assert t.numel() == numel, '%s has %d entries, expected %d' % (name, t.numel(), numel)
    return t.contiguous()


class SparseStorage(object):
    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None,
                 col: Optional[Tensor] = None, value: Optional[Tensor] = None,
                 sparse_sizes: Optional[Tuple[Optional[int], Optional[int]]] = None,
                 rowcount: Optional[Tensor] = None, colptr: Optional[Tensor] = None,
                 colcount: Optional[Tensor] = None, csr2csc: Optional[Tensor] = None,
                 csc2csr: Optional[Tensor] = None, is_sorted: bool = False,
                 trust_data: bool = False):
        assert row is not None or rowptr is not None
        assert col is not None
        col = _check_index(col, 'col', col)
        nnz = col.numel()

        # sizes: given, or inferred (inference reads a maximum back from the device)
        M = N = None
        if sparse_sizes is not None:
            M, N = sparse_sizes
        if M is None:
            if rowptr is not None:
                M = rowptr.numel() - 1
            else:
                M = int(row.max()) + 1 if row.numel() > 0 else 0
        elif rowptr is not None:
            assert rowptr.numel() - 1 == M
        elif not trust_data and row.numel() > 0:
            assert int(row.max()) < M
        if N is None:
            N = int(col.max()) + 1 if nnz > 0 else 0
        elif not trust_data and nnz > 0:
            assert int(col.max()) < N
        self._sparse_sizes = (int(M), int(N))

        self._row = _check_index(row, 'row', col, nnz)
        self._rowptr = _check_index(rowptr, 'rowptr', col, M + 1)
        self._col = col
        if value is not None:
            assert value.device == col.device
            assert value.size(0) == nnz
            value = value.contiguous()
        self._value = value
        self._cache = {
            'rowcount': _check_index(rowcount, 'rowcount', col, M),
            'colptr': _check_index(colptr, 'colptr', col, N + 1),
            'colcount': _check_index(colcount, 'colcount', col, N),
            'csr2csc': _check_index(csr2csc, 'csr2csc', col, nnz),
            'csc2csr': _check_index(csc2csr, 'csc2csr', col, nnz),
        }

        # Unsorted COO handed over on the CPU: there is no CPU sort in this package, so the sort is
        # deferred until the storage is moved to the GPU (`.cuda()` / `.to(device)`); until then every
        # accessor refuses to hand out the (still unsorted) arrays.
        self._pending_sort = bool(not is_sorted and nnz > 1 and not col.is_cuda)
        if not is_sorted and nnz > 1 and col.is_cuda:
            r = self.row()
            descents = int(torch.ops.tsamd.coo_order(r, col, N)[0])  # the one host sync
            if descents > 0:
                rs, cs, perm = torch.ops.tsamd.sort_coo(r, col, M, N, True)
                self._row, self._col = rs, cs
                self._rowptr = None
                if value is not None:
                    self._value = value[perm]
                self._cache['csr2csc'] = None
                self._cache['csc2csr'] = None

    # ---- construction helpers --------------------------------------------------------------
    @classmethod
    def empty(cls):
        z = torch.tensor([], dtype=torch.long)
        return cls(row=z, col=z, sparse_sizes=(0, 0), is_sorted=True, trust_data=True)

    def _derive(self, **overrides):
        """A new storage sharing this one's tensors, with some fields replaced."""
        self._ready()
        kw = dict(row=self._row, rowptr=self._rowptr, col=self._col, value=self._value,
                  sparse_sizes=self._sparse_sizes, is_sorted=True, trust_data=True)
        kw.update(self._cache)
        kw.update(overrides)
        return SparseStorage(**kw)

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
        if self._row is None:
            if self._rowptr is None:
                raise ValueError
            self._row = torch.ops.torch_sparse.ptr2ind(self._rowptr, self._col.numel())
        return self._row

    def has_rowptr(self) -> bool:
        return self._rowptr is not None

    def rowptr(self) -> Tensor:
        self._ready()
        if self._rowptr is None:
            if self._row is None:
                raise ValueError
            self._rowptr = torch.ops.torch_sparse.ind2ptr(self._row, self._sparse_sizes[0])
        return self._rowptr

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
        return self._derive(value=self._layout_value(value, layout))

    # ---- sizes -----------------------------------------------------------------------------
    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sparse_sizes

    def sparse_size(self, dim: int) -> int:
        return self._sparse_sizes[dim]

    def sparse_resize(self, sparse_sizes: Tuple[int, int]):
        assert len(sparse_sizes) == 2
        nnz = self._col.numel()
        out = {}
        for dim, (ptr_key, cnt_key) in enumerate((('rowptr', 'rowcount'), ('colptr', 'colcount'))):
            diff = sparse_sizes[dim] - self._sparse_sizes[dim]
            ptr = self._rowptr if dim == 0 else self._cache['colptr']
            cnt = self._cache[cnt_key]
            if diff > 0:
                if ptr is not None:
                    ptr = torch.cat([ptr, ptr.new_full((diff, ), nnz)])
                if cnt is not None:
                    cnt = torch.cat([cnt, cnt.new_zeros(diff)])
            elif diff < 0:
                ptr = ptr[:diff] if ptr is not None else None
                cnt = cnt[:diff] if cnt is not None else None
            out[ptr_key], out[cnt_key] = ptr, cnt
        return self._derive(sparse_sizes=tuple(sparse_sizes), **out)

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
        return SparseStorage(row=row, col=col, value=self._value, sparse_sizes=(num_rows, num_cols),
                             is_sorted=True, trust_data=True)

    # ---- cached CSC-side views -------------------------------------------------------------
    def has_rowcount(self) -> bool:
        return self._cache['rowcount'] is not None

    def rowcount(self) -> Tensor:
        if self._cache['rowcount'] is None:
            rowptr = self.rowptr()
            self._cache['rowcount'] = rowptr[1:] - rowptr[:-1]
        return self._cache['rowcount']

    def has_csr2csc(self) -> bool:
        return self._cache['csr2csc'] is not None

    def csr2csc(self) -> Tensor:
        """Permutation that orders the entries column-major (stable radix sort of col*M+row)."""
        if self._cache['csr2csc'] is None:
            M, N = self._sparse_sizes
            _, _, perm = torch.ops.tsamd.sort_coo(self._col, self.row(), N, M, False)
            self._cache['csr2csc'] = perm
        return self._cache['csr2csc']

    def has_csc2csr(self) -> bool:
        return self._cache['csc2csr'] is not None

    def csc2csr(self) -> Tensor:
        if self._cache['csc2csr'] is None:
            perm = self.csr2csc()
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), device=perm.device)
            self._cache['csc2csr'] = inv
        return self._cache['csc2csr']

    def has_colptr(self) -> bool:
        return self._cache['colptr'] is not None

    def colptr(self) -> Tensor:
        if self._cache['colptr'] is None:
            N = self._sparse_sizes[1]
            if self._cache['colcount'] is not None:
                ptr = self._col.new_zeros(N + 1)
                torch.cumsum(self._cache['colcount'], dim=0, out=ptr[1:])
            else:
                ptr = torch.ops.torch_sparse.ind2ptr(self._col[self.csr2csc()], N)
            self._cache['colptr'] = ptr
        return self._cache['colptr']

    def has_colcount(self) -> bool:
        return self._cache['colcount'] is not None

    def colcount(self) -> Tensor:
        if self._cache['colcount'] is None:
            ptr = self.colptr()
            self._cache['colcount'] = ptr[1:] - ptr[:-1]
        return self._cache['colcount']

    # ---- coalescing ------------------------------------------------------------------------
    def is_coalesced(self) -> bool:
        self._ready()
        if self._col.numel() <= 1:
            return True
        counts = torch.ops.tsamd.coo_order(self.row(), self._col, self._sparse_sizes[1]).tolist()
        return counts[0] == 0 and counts[1] == 0

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
            from .segment import segment_reduce
            # heavy duplication (few distinct pairs, long runs): take the entry-balanced path
            value = segment_reduce(value, None, seg_ptr, n, reduce, balanced=nnz > 8 * max(n, 1))
        return SparseStorage(row=row_u[:n].clone(), col=col_u[:n].clone(), value=value,
                             sparse_sizes=self._sparse_sizes, is_sorted=True, trust_data=True)

    # ---- cache management ------------------------------------------------------------------
    def fill_cache_(self):
        self.row(), self.rowptr(), self.rowcount()
        self.csr2csc(), self.csc2csr(), self.colptr(), self.colcount()
        return self

    def clear_cache_(self):
        for k in _CACHE_KEYS:
            self._cache[k] = None
        return self

    def cached_keys(self) -> List[str]:
        return [k for k in _CACHE_KEYS if self._cache[k] is not None]

    def num_cached_keys(self) -> int:
        return len(self.cached_keys())

    # ---- copies / device / dtype -----------------------------------------------------------
    def _map(self, fn, value_fn=None):
        def ap(t, f):
            return None if t is None else f(t)
        return SparseStorage(row=ap(self._row, fn), rowptr=ap(self._rowptr, fn), col=fn(self._col),
                             value=ap(self._value, value_fn or fn), sparse_sizes=self._sparse_sizes,
                             is_sorted=not self._pending_sort, trust_data=True,
                             **{k: ap(v, fn) for k, v in self._cache.items()})

    def copy(self):
        return self._derive()

    def clone(self):
        return self._map(lambda t: t.clone())

    def type(self, dtype: torch.dtype, non_blocking: bool = False):
        if self._value is None or dtype == self._value.dtype:
            return self
        return self.set_value(self._value.to(dtype=dtype, non_blocking=non_blocking), layout='coo')

    def type_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.type(tensor.dtype, non_blocking)

    def to_device(self, device: torch.device, non_blocking: bool = False):
        if device == self._col.device:
            return self
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def device_as(self, tensor: Tensor, non_blocking: bool = False):
        return self.to_device(tensor.device, non_blocking)

    def cuda(self):
        return self if self._col.is_cuda else self._map(lambda t: t.cuda())

    def cpu(self):
        return self._map(lambda t: t.cpu()) if self._col.is_cuda else self

    def is_cuda(self) -> bool:
        return self._col.is_cuda

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())

    def is_pinned(self) -> bool:
        ts = [self._row, self._rowptr, self._col, self._value] + list(self._cache.values())
        return all(t.is_pinned() for t in ts if t is not None)

    def share_memory_(self):
        for t in [self._row, self._rowptr, self._col, self._value] + list(self._cache.values()):
            if t is not None:
                t.share_memory_()
        return self

    def is_shared(self) -> bool:
        ts = [self._row, self._rowptr, self._col, self._value] + list(self._cache.values())
        return all(t.is_shared() for t in ts if t is not None)

    # private attribute names the reference exposes and its front-ends read directly
    @property
    def _rowcount(self):
        return self._cache['rowcount']

    @property
    def _colptr(self):
        return self._cache['colptr']

    @property
    def _colcount(self):
        return self._cache['colcount']

    @property
    def _csr2csc(self):
        return self._cache['csr2csc']

    @property
    def _csc2csr(self):
        return self._cache['csc2csr']

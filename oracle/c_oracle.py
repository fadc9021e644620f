"""numpy front-end of ts_oracle.c (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Arrays are numpy; f16 is ``np.float16``; bf16 is carried as ``np.uint16`` bit patterns
(helpers ``f32_to_bf16_bits`` / ``bf16_bits_to_f32`` below).
"""
import ctypes

import numpy as np

from . import build

F32, F64, F16, BF16, I32, I64, U8, I8, I16 = range(9)
SUM, MEAN, MIN, MAX = range(4)
REDUCE = {'sum': SUM, 'add': SUM, 'mean': MEAN, 'min': MIN, 'max': MAX}
NP_DTYPE = {F32: np.float32, F64: np.float64, F16: np.float16, BF16: np.uint16, I32: np.int32,
            I64: np.int64, U8: np.uint8, I8: np.int8, I16: np.int16}

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    r[np.isnan(x)] = 0x7FC0
    return r


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def spmm(dtype, reduce, rowptr, col, value, mat, wide_acc=False):
    """Returns (out, arg_out or None). mat: [..., N, K]."""
    red = REDUCE[reduce] if isinstance(reduce, str) else reduce
    rowptr, col = _i64(rowptr), _i64(col)
    npdt = NP_DTYPE[dtype]
    mat = np.ascontiguousarray(mat, dtype=npdt)
    assert mat.ndim >= 2
    if value is not None:
        value = np.ascontiguousarray(value, dtype=npdt)
    M, E = rowptr.size - 1, col.size
    N, K = mat.shape[-2], mat.shape[-1]
    B = int(np.prod(mat.shape[:-2])) if mat.ndim > 2 else 1
    oshape = mat.shape[:-2] + (M, K)
    out = np.empty(oshape, dtype=npdt)
    arg = np.empty(oshape, dtype=np.int64) if red in (MIN, MAX) else None
    rc = lib().ts_oracle_spmm(dtype, red, _p(rowptr), _p(col), _p(value), _p(mat), _p(out), _p(arg),
                              ctypes.c_int64(B), ctypes.c_int64(M), ctypes.c_int64(N),
                              ctypes.c_int64(K), ctypes.c_int64(E), int(bool(wide_acc)))
    assert rc == 0
    return out, arg


def spmm_value_bw(dtype, reduce, row, rowptr, col, mat, grad, wide_acc=False):
    red = REDUCE[reduce] if isinstance(reduce, str) else reduce
    row, rowptr, col = _i64(row), _i64(rowptr), _i64(col)
    npdt = NP_DTYPE[dtype]
    mat = np.ascontiguousarray(mat, dtype=npdt)
    grad = np.ascontiguousarray(grad, dtype=npdt)
    M, E = rowptr.size - 1, col.size
    N, K = mat.shape[-2], mat.shape[-1]
    B = int(np.prod(mat.shape[:-2])) if mat.ndim > 2 else 1
    out = np.empty((E, ), dtype=npdt)
    rc = lib().ts_oracle_spmm_value_bw(dtype, red, _p(row), _p(rowptr), _p(col), _p(mat), _p(grad),
                                       _p(out), ctypes.c_int64(B), ctypes.c_int64(M),
                                       ctypes.c_int64(N), ctypes.c_int64(K), ctypes.c_int64(E),
                                       int(bool(wide_acc)))
    assert rc == 0
    return out


def spmm_minmax_bw(dtype, col, value, mat, grad_out, arg_out, want_value=True, want_mat=True):
    col = _i64(col)
    npdt = NP_DTYPE[dtype]
    mat = np.ascontiguousarray(mat, dtype=npdt)
    grad_out = np.ascontiguousarray(grad_out, dtype=npdt)
    arg_out = _i64(arg_out)
    if value is not None:
        value = np.ascontiguousarray(value, dtype=npdt)
    E = col.size
    N, K = mat.shape[-2], mat.shape[-1]
    M = grad_out.shape[-2]
    B = int(np.prod(mat.shape[:-2])) if mat.ndim > 2 else 1
    gv = np.empty((E, ), dtype=npdt) if want_value else None
    gm = np.empty(mat.shape, dtype=npdt) if want_mat else None
    rc = lib().ts_oracle_spmm_minmax_bw(dtype, _p(col), _p(value), _p(mat), _p(grad_out),
                                        _p(arg_out), _p(gv), _p(gm), ctypes.c_int64(B),
                                        ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K),
                                        ctypes.c_int64(E))
    assert rc == 0
    return gv, gm


def ind2ptr(ind, M):
    ind = _i64(ind)
    out = np.empty((M + 1, ), dtype=np.int64)
    lib().ts_oracle_ind2ptr(_p(ind), ctypes.c_int64(M), ctypes.c_int64(ind.size), _p(out))
    return out


def ptr2ind(ptr, E):
    ptr = _i64(ptr)
    out = np.empty((E, ), dtype=np.int64)
    lib().ts_oracle_ptr2ind(_p(ptr), ctypes.c_int64(ptr.size - 1), ctypes.c_int64(E), _p(out))
    return out

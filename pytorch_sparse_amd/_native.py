"""ctypes binding of the C-ABI in ``include/tsamd.h`` (``lib/libtsamd.so``).

This is the thin layer that hands torch device pointers, sizes and the current
HIP stream to the hand-written kernels.  There is no CPU implementation behind
it: a missing library, a CPU tensor or a non-zero status raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TSAMD_LIB') or os.path.join(_HERE, 'lib', 'libtsamd.so')

# Every symbol include/tsamd.h declares (tests check that all of them resolve).
SYMBOLS = [
    'tsamd_hip_version', 'tsamd_build_flags', 'tsamd_last_hip_error', 'tsamd_status_string',
    'tsamd_spmm_workspace_bytes', 'tsamd_spmm', 'tsamd_spmm_reference_order', 'tsamd_spmm_permuted', 'tsamd_spmm_profiled',
    'tsamd_spmm_partial_workspace_bytes', 'tsamd_spmm_partial',
    'tsamd_spmm_operand_cache_bytes', 'tsamd_spmm_cached_workspace_bytes', 'tsamd_spmm_cached',
    'tsamd_spmm_minmax_arg32', 'tsamd_spmm_minmax_bw_csc_arg32',
    'tsamd_spmm_minmax_records_in_forward', 'tsamd_spmm_minmax_records_bytes', 'tsamd_spmm_minmax_records_workspace_bytes', 'tsamd_spmm_minmax_records',
    'tsamd_spmm_minmax_winrec', 'tsamd_spmm_minmax_bw_csc_records_workspace_bytes', 'tsamd_spmm_minmax_bw_csc_records',
    'tsamd_gather_rows', 'tsamd_relabel_ids', 'tsamd_spmm_relabelled_workspace_bytes', 'tsamd_spmm_relabelled',
    'tsamd_spmm_coo_small_supported', 'tsamd_spmm_coo_small',
    'tsamd_spmm_value_bw',
    'tsamd_spmm_minmax_bw_workspace_bytes', 'tsamd_spmm_minmax_bw',
    'tsamd_spmm_minmax_bw_csc_workspace_bytes', 'tsamd_spmm_minmax_bw_csc',
    'tsamd_ind2ptr', 'tsamd_ptr2ind',
    'tsamd_coo_order', 'tsamd_coo_check', 'tsamd_sort_rank_mode', 'tsamd_sort_coalesce_workspace_bytes', 'tsamd_sort_coalesce', 'tsamd_sort_coalesce_reduce', 'tsamd_sort_coo_workspace_bytes', 'tsamd_sort_coo', 'tsamd_sort_coo_auto', 'tsamd_sort_coo_probed', 'tsamd_sort_coo_values',
    'tsamd_coalesce_workspace_bytes', 'tsamd_coalesce_index', 'tsamd_segment_reduce',
    'tsamd_segment_reduce_balanced_workspace_bytes', 'tsamd_segment_reduce_balanced',
    'tsamd_exclusive_scan_workspace_bytes', 'tsamd_exclusive_scan_i64',
    'tsamd_spspmm_plan', 'tsamd_spspmm_workspace_bytes', 'tsamd_spspmm_symbolic', 'tsamd_spspmm_numeric',
    'tsamd_select_workspace_bytes', 'tsamd_select_plan', 'tsamd_select_fill',
    'tsamd_filter_workspace_bytes', 'tsamd_filter_plan', 'tsamd_filter_apply',
    'tsamd_filter_tiles_workspace_bytes', 'tsamd_filter_count', 'tsamd_filter_write',
    'tsamd_scatter_rows',
    'tsamd_num_diag', 'tsamd_non_diag_mask', 'tsamd_insert_diag', 'tsamd_set_diag_apply',
    'tsamd_random_walk', 'tsamd_sample_workspace_bytes', 'tsamd_sample_plan', 'tsamd_sample_draw',
    'tsamd_relabel_workspace_bytes', 'tsamd_relabel_plan', 'tsamd_relabel_apply', 'tsamd_relabel_seed',
    'tsamd_relabel_extend', 'tsamd_temporal_mark', 'tsamd_temporal_redraw_workspace_bytes', 'tsamd_temporal_redraw',
    'tsamd_temporal_relabel_workspace_bytes', 'tsamd_temporal_relabel', 'tsamd_temporal_emit', 'tsamd_subset_assoc',
]

DTYPES = {
    torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3,
    torch.int32: 4, torch.int64: 5, torch.uint8: 6, torch.int8: 7, torch.int16: 8,
}
REDUCES = {'sum': 0, 'add': 0, 'mean': 1, 'min': 2, 'max': 3}

_lib = None


class TsamdError(RuntimeError):
    pass


def lib():
    """Load libtsamd.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'pytorch_sparse_amd: %s is missing. Build it with '
                '`python pytorch_sparse_amd/build.py` (needs hipcc, targets gfx950). '
                'There is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.tsamd_hip_version.restype = ctypes.c_int64
        L.tsamd_status_string.restype = ctypes.c_char_p
        L.tsamd_spmm_workspace_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_partial_workspace_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_minmax_bw_workspace_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_minmax_bw_csc_workspace_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_minmax_records_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_minmax_records_workspace_bytes.restype = ctypes.c_size_t
        L.tsamd_spmm_minmax_bw_csc_records_workspace_bytes.restype = ctypes.c_size_t
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        msg = L.tsamd_status_string(int(status)).decode()
        if status == 3:
            msg += ' (hipError_t %d)' % L.tsamd_last_hip_error()
        raise TsamdError('%s failed: %s' % (what, msg))


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _i64(x):
    return ctypes.c_int64(int(x))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TsamdError(
                'pytorch_sparse_amd runs on MI355X (HIP) tensors only; got a %s tensor. '
                'There is no CPU implementation in this package.' % t.device.type)


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dtype):
    try:
        return DTYPES[dtype]
    except KeyError:
        raise TsamdError('unsupported dtype %s (supported: %s)' % (dtype, sorted(map(str, DTYPES))))


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def spmm(rowptr, col, value, mat, reduce, out=None, profile=None):
    """C-ABI ``tsamd_spmm``: returns (out, arg_out or None).  mat: [..., N, K] contiguous.
    ``profile``: optional list; receives [partition_ms, merge_ms, fixup_ms] (synchronises)."""
    require_gpu(rowptr, col, value, mat)
    red = REDUCES[reduce]
    dt = dtype_code(mat.dtype)
    if value is not None and value.dtype != mat.dtype:
        raise TsamdError('expected scalar type %s but found %s' % (mat.dtype, value.dtype))
    if mat.dim() < 2:
        raise TsamdError('Input mismatch: mat.dim() >= 2 required')
    mat = mat.contiguous()
    M, E = rowptr.numel() - 1, col.numel()
    N, K = mat.size(-2), mat.size(-1)
    B = mat.numel() // max(N * K, 1) if N * K > 0 else 1
    sizes = list(mat.shape)
    sizes[-2] = M
    if out is None:
        out = torch.empty(sizes, dtype=mat.dtype, device=mat.device)
    arg = None
    if red >= 2:
        arg = torch.empty(sizes, dtype=torch.int64, device=mat.device)
    L = lib()
    nb = L.tsamd_spmm_workspace_bytes(dt, red, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E))
    ws = workspace(nb, mat.device)
    with torch.cuda.device(mat.device):
        args = (dt, red, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(out), _ptr(arg),
                _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(ws), ctypes.c_size_t(ws.numel()),
                stream_ptr(mat.device))
        if profile is None:
            st = L.tsamd_spmm(*args)
        else:
            ms = (ctypes.c_float * 3)()
            st = L.tsamd_spmm_profiled(*args, ms)
            profile[:] = [float(ms[0]), float(ms[1]), float(ms[2])]
    check(st, 'tsamd_spmm')
    return out, arg


def spmm_partial(rowptr, col, value, mat, reduce, out, arg_out=None, arg_map=None, arg_none=0, accumulate=True,
                 deg_rowptr=None):
    """C-ABI ``tsamd_spmm_partial``: the product of one column block combined into ``out`` / ``arg_out`` in place
    (``accumulate=False``: the first block overwrites them).  mat: [N, K] or [B, N, K] contiguous."""
    require_gpu(rowptr, col, value, mat, out, arg_out, arg_map, deg_rowptr)
    red = REDUCES[reduce]
    dt = dtype_code(mat.dtype)
    if value is not None and value.dtype != mat.dtype:
        raise TsamdError('expected scalar type %s but found %s' % (mat.dtype, value.dtype))
    if not (mat.is_contiguous() and out.is_contiguous() and (arg_out is None or arg_out.is_contiguous())):
        raise TsamdError('tsamd_spmm_partial works in place: mat / out / arg_out must be contiguous')
    if out.dtype != mat.dtype or (red >= 2 and (arg_out is None or arg_out.dtype != torch.int64)):
        raise TsamdError('Input mismatch: out / arg_out')
    M, E = rowptr.numel() - 1, col.numel()
    N, K = mat.size(-2), mat.size(-1)
    B = mat.numel() // max(N * K, 1) if N * K > 0 else 1
    if out.numel() != B * M * K or (arg_out is not None and arg_out.numel() != out.numel()):
        raise TsamdError('Input mismatch: out must be [B, M, K]')
    L = lib()
    nb = L.tsamd_spmm_partial_workspace_bytes(dt, red, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E))
    ws = workspace(nb, mat.device)
    with torch.cuda.device(mat.device):
        st = L.tsamd_spmm_partial(dt, red, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(out), _ptr(arg_out),
                                  _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(arg_map), _i64(arg_none),
                                  ctypes.c_int(1 if accumulate else 0), _ptr(deg_rowptr), _ptr(ws),
                                  ctypes.c_size_t(ws.numel()), stream_ptr(mat.device))
    check(st, 'tsamd_spmm_partial')
    return out, arg_out


def spmm_value_bw(row, rowptr, col, mat, grad, reduce):
    """C-ABI ``tsamd_spmm_value_bw``: gradient w.r.t. the sparse values, [E]."""
    require_gpu(rowptr, col, mat, grad, row)
    red = REDUCES[reduce]
    dt = dtype_code(mat.dtype)
    mat, grad = mat.contiguous(), grad.contiguous()
    if grad.dtype != mat.dtype:
        raise TsamdError('expected scalar type %s but found %s' % (mat.dtype, grad.dtype))
    M, E = rowptr.numel() - 1, col.numel()
    N, K = mat.size(-2), mat.size(-1)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    out = torch.empty(E, dtype=mat.dtype, device=mat.device)
    with torch.cuda.device(mat.device):
        st = lib().tsamd_spmm_value_bw(dt, red, _ptr(row), _ptr(rowptr), _ptr(col), _ptr(mat),
                                       _ptr(grad), _ptr(out), _i64(B), _i64(M), _i64(N), _i64(K),
                                       _i64(E), stream_ptr(mat.device))
    check(st, 'tsamd_spmm_value_bw')
    return out


def spmm_minmax_bw(rowptr, col, value, mat, grad_out, arg_out, want_value=True, want_mat=True):
    """C-ABI ``tsamd_spmm_minmax_bw``: (grad_value or None, grad_mat or None).  ``rowptr`` may be
    None when ``want_value`` is False."""
    require_gpu(rowptr, col, value, mat, grad_out, arg_out)
    dt = dtype_code(mat.dtype)
    mat, grad_out, arg_out = mat.contiguous(), grad_out.contiguous(), arg_out.contiguous()
    E = col.numel()
    N, K = mat.size(-2), mat.size(-1)
    M = grad_out.size(-2)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    gv = torch.empty(E, dtype=mat.dtype, device=mat.device) if want_value else None
    gm = torch.empty_like(mat) if want_mat else None
    L = lib()
    nb = L.tsamd_spmm_minmax_bw_workspace_bytes(dt, _i64(B), _i64(N), _i64(K), _i64(E))
    ws = workspace(nb, mat.device)
    with torch.cuda.device(mat.device):
        st = L.tsamd_spmm_minmax_bw(dt, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(grad_out),
                                    _ptr(arg_out), _ptr(gv), _ptr(gm), _i64(B), _i64(M), _i64(N),
                                    _i64(K), _i64(E), _ptr(ws), ctypes.c_size_t(ws.numel()),
                                    stream_ptr(mat.device))
    check(st, 'tsamd_spmm_minmax_bw')
    return gv, gm


def spmm_minmax_arg32(rowptr, col, value, mat, reduce):
    """C-ABI ``tsamd_spmm_minmax_arg32`` (stateless): min / max with the winners as int32 ids -> (out, arg32)."""
    require_gpu(rowptr, col, value, mat)
    mat = mat.contiguous()
    M, E = rowptr.numel() - 1, col.numel()
    N, K = mat.size(-2), mat.size(-1)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    red = REDUCES[reduce]
    dt = dtype_code(mat.dtype)
    out = torch.empty(list(mat.shape[:-2]) + [M, K], dtype=mat.dtype, device=mat.device)
    arg = torch.empty(out.shape, dtype=torch.int32, device=mat.device)
    L = lib()
    nbytes = L.tsamd_spmm_workspace_bytes(dt, red, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E))
    ws = workspace(nbytes, mat.device)
    with torch.cuda.device(mat.device):
        st = L.tsamd_spmm_minmax_arg32(dt, red, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(out), _ptr(arg),
                                       _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(ws),
                                       ctypes.c_size_t(ws.numel()), None, ctypes.c_size_t(0), 0,
                                       stream_ptr(mat.device))
    check(st, 'tsamd_spmm_minmax_arg32')
    return out, arg


def spmm_minmax_bw_csc(rowptr, col, value, mat, grad_out, arg_out, colptr, csr2csc, row, want_value=True,
                       want_mat=True):
    """C-ABI ``tsamd_spmm_minmax_bw_csc`` (pull formulation over the CSC arrays, no atomics):
    (grad_value or None, grad_mat or None)."""
    require_gpu(rowptr, col, value, mat, grad_out, arg_out, colptr, csr2csc, row)
    dt = dtype_code(mat.dtype)
    mat, grad_out, arg_out = mat.contiguous(), grad_out.contiguous(), arg_out.contiguous()
    E = col.numel()
    N, K = mat.size(-2), mat.size(-1)
    M = grad_out.size(-2)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    gv = torch.empty(E, dtype=mat.dtype, device=mat.device) if want_value else None
    gm = torch.empty_like(mat) if want_mat else None
    L = lib()
    nb = L.tsamd_spmm_minmax_bw_csc_workspace_bytes(dt, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E))
    ws = workspace(nb, mat.device)
    fn = L.tsamd_spmm_minmax_bw_csc_arg32 if arg_out.dtype == torch.int32 else L.tsamd_spmm_minmax_bw_csc
    with torch.cuda.device(mat.device):
        st = fn(dt, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(grad_out),
                _ptr(arg_out), _ptr(colptr), _ptr(csr2csc), _ptr(row), _ptr(gv), _ptr(gm),
                _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(ws),
                ctypes.c_size_t(ws.numel()), stream_ptr(mat.device))
    check(st, 'tsamd_spmm_minmax_bw_csc')
    return gv, gm


def spmm_minmax_records(rowptr, col, value, mat, reduce, row, zero=False):
    """C-ABI ``tsamd_spmm_minmax_records``: min / max forward that leaves the winner records of the pull backward
    -> (out, records as int32 words).  zero: the buffer starts as zeros (padding words of a record are never written)."""
    require_gpu(rowptr, col, value, mat, row)
    mat = mat.contiguous()
    M, E = rowptr.numel() - 1, col.numel()
    N, K = mat.size(-2), mat.size(-1)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    red = REDUCES[reduce]
    dt = dtype_code(mat.dtype)
    out = torch.empty(list(mat.shape[:-2]) + [M, K], dtype=mat.dtype, device=mat.device)
    L = lib()
    rec = (torch.zeros if zero else torch.empty)(L.tsamd_spmm_minmax_records_bytes(_i64(B), _i64(K), _i64(E)) // 4,
                                                 dtype=torch.int32, device=mat.device)
    ws = workspace(L.tsamd_spmm_minmax_records_workspace_bytes(dt, red, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E)),
                   mat.device)
    with torch.cuda.device(mat.device):
        st = L.tsamd_spmm_minmax_records(dt, red, _ptr(rowptr), _ptr(col), _ptr(value), _ptr(mat), _ptr(out), _ptr(row),
                                         _ptr(rec), _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(ws),
                                         ctypes.c_size_t(ws.numel()), stream_ptr(mat.device))
    check(st, 'tsamd_spmm_minmax_records')
    return out, rec


def spmm_minmax_winrec(row, value, arg32, K):
    """C-ABI ``tsamd_spmm_minmax_winrec``: the winner records of int32 winner ids [B, M, K] -> int32 words."""
    require_gpu(row, value, arg32)
    arg32 = arg32.contiguous()
    E, M = row.numel(), arg32.size(-2)
    B = arg32.numel() // (M * K) if M * K > 0 else 1
    L = lib()
    rec = torch.zeros(L.tsamd_spmm_minmax_records_bytes(_i64(B), _i64(K), _i64(E)) // 4, dtype=torch.int32, device=row.device)
    dt = dtype_code(value.dtype) if value is not None else 0
    with torch.cuda.device(row.device):
        st = L.tsamd_spmm_minmax_winrec(dt, _ptr(row), _ptr(value), _ptr(arg32), _ptr(rec), _i64(B), _i64(M), _i64(K),
                                        _i64(E), stream_ptr(row.device))
    check(st, 'tsamd_spmm_minmax_winrec')
    return rec


def spmm_minmax_bw_csc_records(rowptr, col, has_value, mat, grad_out, records, colptr, csr2csc, row, want_value=False):
    """C-ABI ``tsamd_spmm_minmax_bw_csc_records``: the pull backward on records -> (grad_value or None, grad_mat)."""
    require_gpu(rowptr, col, mat, grad_out, records, colptr, csr2csc, row)
    dt = dtype_code(mat.dtype)
    mat, grad_out = mat.contiguous(), grad_out.contiguous()
    E = col.numel()
    N, K = mat.size(-2), mat.size(-1)
    M = grad_out.size(-2)
    B = mat.numel() // (N * K) if N * K > 0 else 1
    gv = torch.empty(E, dtype=mat.dtype, device=mat.device) if want_value else None
    gm = torch.empty_like(mat)
    L = lib()
    ws = workspace(L.tsamd_spmm_minmax_bw_csc_records_workspace_bytes(dt, _i64(B), _i64(M), _i64(N), _i64(K), _i64(E)),
                   mat.device)
    with torch.cuda.device(mat.device):
        st = L.tsamd_spmm_minmax_bw_csc_records(dt, _ptr(rowptr), _ptr(col), ctypes.c_int(1 if has_value else 0), _ptr(mat),
                                                _ptr(grad_out), _ptr(records), _ptr(colptr), _ptr(csr2csc), _ptr(row),
                                                _ptr(gv), _ptr(gm), _i64(B), _i64(M), _i64(N), _i64(K), _i64(E), _ptr(ws),
                                                ctypes.c_size_t(ws.numel()), stream_ptr(mat.device))
    check(st, 'tsamd_spmm_minmax_bw_csc_records')
    return gv, gm


def ind2ptr(ind, M):
    require_gpu(ind)
    out = torch.empty(M + 1, dtype=torch.int64, device=ind.device)
    with torch.cuda.device(ind.device):
        st = lib().tsamd_ind2ptr(_ptr(ind), _i64(M), _i64(ind.numel()), _ptr(out),
                                 stream_ptr(ind.device))
    check(st, 'tsamd_ind2ptr')
    return out


def ptr2ind(ptr, E):
    require_gpu(ptr)
    out = torch.empty(E, dtype=torch.int64, device=ptr.device)
    with torch.cuda.device(ptr.device):
        st = lib().tsamd_ptr2ind(_ptr(ptr), _i64(ptr.numel() - 1), _i64(E), _ptr(out),
                                 stream_ptr(ptr.device))
    check(st, 'tsamd_ptr2ind')
    return out

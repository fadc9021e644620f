"""numpy restatement of the reference's Python-level pipelines -- TEST INFRASTRUCTURE ONLY.

  sort_coo / csr2csc      torch_sparse/storage.py:149-162, 407-416 (key = row * N + col, sort)
  coalesce                torch_sparse/storage.py:436-466 + torch_scatter.segment_csr semantics
  transpose               torch_sparse/transpose.py:39-62
  spspmm                  torch_sparse/matmul.py:94-111 -> torch.sparse.mm; PyTorch's SpGEMM is not
                          under /root/reference (third-party: torch, unpinned by the reference,
                          2.10.0 here), so this restates the published row-wise (Gustavson)
                          definition: C[i, :] = sum_k A[i, k] * B[k, :], rows sorted, duplicates summed.

Pinned by tests/test_oracle.py against the fixtures written by tests/golden/make_golden.py (which
ran the reference's own Python) -- see that file.  The reference's sort is not stable; this one
is, which only matters for the order in which duplicate values are reduced.
"""
import numpy as np


def sort_coo(row, col, m, n):
    key = np.asarray(row, dtype=np.int64) * n + np.asarray(col, dtype=np.int64)
    perm = np.argsort(key, kind='stable')
    return np.asarray(row)[perm], np.asarray(col)[perm], perm


def csr2csc(row, col, m, n):
    key = np.asarray(col, dtype=np.int64) * m + np.asarray(row, dtype=np.int64)
    return np.argsort(key, kind='stable')


def coalesce(row, col, value, m, n, op='add'):
    row, col, perm = sort_coo(row, col, m, n)
    key = row.astype(np.int64) * n + col
    head = np.ones(key.shape, dtype=bool)
    head[1:] = key[1:] != key[:-1]
    starts = np.nonzero(head)[0]
    out_row, out_col = row[head], col[head]
    if value is None:
        return out_row, out_col, None
    v = np.asarray(value)[perm]
    if starts.size == 0:
        return out_row, out_col, v[:0]
    if op in ('add', 'sum'):
        out = np.add.reduceat(v, starts, axis=0)
    elif op == 'mean':
        cnt = np.diff(np.append(starts, key.size))
        s = np.add.reduceat(v, starts, axis=0)
        cnt = cnt.reshape((-1, ) + (1, ) * (v.ndim - 1))
        out = np.floor_divide(s, cnt) if np.issubdtype(v.dtype, np.integer) else s / cnt
    elif op == 'min':
        out = np.minimum.reduceat(v, starts, axis=0)
    elif op == 'max':
        out = np.maximum.reduceat(v, starts, axis=0)
    else:
        raise ValueError(op)
    return out_row, out_col, out.astype(v.dtype)


def transpose(row, col, value, m, n):
    return coalesce(col, row, value, n, m, 'add')


def spspmm(rowA, colA, valA, rowB, colB, valB, m, k, n):
    """Row-wise SpGEMM on coalesced COO inputs; returns sorted (row, col, val)."""
    rowA, colA, rowB, colB = (np.asarray(x, dtype=np.int64) for x in (rowA, colA, rowB, colB))
    valA = np.ones(rowA.size) if valA is None else np.asarray(valA)
    valB = np.ones(rowB.size) if valB is None else np.asarray(valB)
    ptrB = np.zeros(k + 1, dtype=np.int64)
    np.add.at(ptrB, rowB + 1, 1)
    ptrB = np.cumsum(ptrB)
    cnt = ptrB[colA + 1] - ptrB[colA]
    total = int(cnt.sum())
    ea = np.repeat(np.arange(rowA.size), cnt)
    off = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    eb = ptrB[colA[ea]] + off
    prow, pcol = rowA[ea], colB[eb]
    pval = (valA[ea] * valB[eb]).astype(np.result_type(valA.dtype, valB.dtype))
    return coalesce(prow, pcol, pval, m, n, 'add')

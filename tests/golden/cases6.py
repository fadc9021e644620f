"""Randomised differential cases shared by the golden generator (reference package, make_golden.py
part 6) and tests/test_random_cases_gpu.py (pytorch_sparse_amd): several hundred small random
matrices -- empty rows / columns, non-square shapes, optional and multi-dimensional values,
non-coalesced inputs -- pushed through one randomly parametrised public-API call each.

A case is a plain dict of numpy arrays / ints / strings (so that it can be stored in one .npz);
`run_case(ts, case, device)` executes it with either package and returns a list of numpy arrays.
"""
import numpy as np
import torch

OPS = ['narrow', 'index_select', 'masked_select', 'remove_diag', 'set_diag', 'fill_diag', 'get_diag', 'cat',
       'permute', 'mul_dense', 'add_dense', 'add_sparse', 'mul_sparse', 'reduce', 'transpose', 'coalesce',
       'masked_select_nnz', 'index_select_nnz', 'to_symmetric', 'getitem', 'saint', 'sample_all', 'eye_matmul',
       'f_spmm', 'f_coalesce', 'f_transpose', 'f_spspmm', 'matmul_grad']


def _rand_matrix(rng, m, n, coalesced=True, vdim=None):
    nnz = int(rng.integers(0, max(1, min(m * n, 60)) + 1))
    if coalesced:
        key = np.unique(rng.integers(0, max(m * n, 1), nnz)) if m * n > 0 else np.zeros(0, np.int64)
    else:
        key = np.sort(rng.integers(0, max(m * n, 1), nnz)) if m * n > 0 else np.zeros(0, np.int64)
    row, col = (key // max(n, 1)).astype(np.int64), (key % max(n, 1)).astype(np.int64)
    if vdim is None:
        val = np.zeros(0, np.float32)
        has = 0
    else:
        shape = (key.size, ) if vdim == 1 else (key.size, vdim)
        val = (rng.integers(-8, 9, shape) / 4).astype(np.float32)
        has = 1
    return dict(row=row, col=col, val=val, has=has, m=m, n=n)


def make_case(seed):
    rng = np.random.default_rng(seed)
    op = OPS[seed % len(OPS)]
    square = op in ('permute', 'saint', 'sample_all', 'to_symmetric', 'eye_matmul') or rng.random() < 0.3
    m = int(rng.integers(1, 14))
    n = m if square else int(rng.integers(1, 14))
    vdim = [None, 1, 1, 3][int(rng.integers(0, 4))]
    if op in ('mul_sparse', 'coalesce', 'reduce', 'eye_matmul') and vdim is None:
        vdim = 1
    if op in ('reduce', 'eye_matmul', 'mul_sparse', 'f_spmm', 'f_spspmm', 'matmul_grad'):
        vdim = 1
    if op in ('f_coalesce', 'f_transpose') and vdim is None:
        vdim = 1
    if op in ('mul_dense', 'add_dense') and vdim == 3:  # the reference's in-place broadcast rejects these
        vdim = 1
    # duplicates only where the reference's result does not depend on an unstable sort
    coalesced = op not in ('coalesce', 'reduce', 'narrow', 'f_spmm', 'f_coalesce') or rng.random() < 0.5
    A = _rand_matrix(rng, m, n, coalesced, vdim)
    c = dict(op=op, seed=seed, **{'A_' + k: v for k, v in A.items()})
    if op == 'narrow':
        dim = int(rng.integers(0, 2))
        size = (m, n)[dim]
        start = int(rng.integers(0, size))
        c.update(dim=dim, start=start, length=int(rng.integers(0, size - start + 1)))
    elif op == 'index_select':
        dim = int(rng.integers(0, 2))
        c.update(dim=dim, idx=rng.integers(0, (m, n)[dim], int(rng.integers(0, 12))).astype(np.int64))
    elif op == 'masked_select':
        dim = int(rng.integers(0, 2))
        c.update(dim=dim, mask=rng.random((m, n)[dim]) < 0.5)
    elif op in ('remove_diag', 'set_diag', 'fill_diag'):
        c.update(k=int(rng.integers(-min(m, 3) + 1, min(n, 3))))
    elif op == 'cat':
        dim = int(rng.integers(0, 3))  # 2 = diagonal
        B = _rand_matrix(rng, m if dim == 1 else int(rng.integers(1, 9)), n if dim == 0 else int(rng.integers(1, 9)),
                         True, vdim)
        c.update(dim=dim, **{'B_' + k: v for k, v in B.items()})
    elif op == 'permute':
        c.update(perm=rng.permutation(m).astype(np.int64))
    elif op in ('mul_dense', 'add_dense'):
        rowwise = int(rng.integers(0, 2))
        c.update(rowwise=rowwise, vec=(rng.integers(-4, 5, m if rowwise else n) / 2).astype(np.float32))
    elif op in ('add_sparse', 'mul_sparse'):
        B = _rand_matrix(rng, int(rng.integers(1, 14)), int(rng.integers(1, 14)), True, vdim)
        c.update(**{'B_' + k: v for k, v in B.items()})
    elif op == 'reduce':
        c.update(dim=int(rng.integers(0, 2)), reduce=['sum', 'mean', 'min', 'max'][int(rng.integers(0, 4))])
    elif op == 'coalesce':
        c.update(reduce=['sum', 'mean', 'min', 'max'][int(rng.integers(0, 4))])
    elif op == 'masked_select_nnz':
        c.update(mask=rng.random(A['row'].size) < 0.5, layout=['coo', 'csc'][int(rng.integers(0, 2))])
    elif op == 'index_select_nnz':
        k = A['row'].size
        c.update(idx=np.sort(rng.choice(k, int(rng.integers(0, k + 1)), replace=False)).astype(np.int64) if k else
                 np.zeros(0, np.int64))
    elif op == 'getitem':
        c.update(r0=int(rng.integers(0, m)), r1=int(rng.integers(0, m + 1)), mask=rng.random(n) < 0.6)
    elif op in ('saint', 'sample_all'):
        c.update(idx=rng.permutation(m)[:int(rng.integers(0, m + 1))].astype(np.int64))
    elif op == 'eye_matmul':
        c.update(x=(rng.integers(-4, 5, (n, 3)) / 2).astype(np.float32),
                 reduce=['sum', 'mean', 'min', 'max'][int(rng.integers(0, 4))])
    elif op in ('f_spmm', 'f_coalesce', 'f_transpose'):
        c.update(shuffle=rng.permutation(A['row'].size).astype(np.int64), x=(rng.integers(-4, 5, (n, 2)) / 2).astype(np.float32),
                 reduce=['add', 'mean', 'min', 'max'][int(rng.integers(0, 4))])
    elif op == 'f_spspmm':
        B = _rand_matrix(rng, n, int(rng.integers(1, 14)), True, 1)
        c.update(**{'B_' + k: v for k, v in B.items()})
    elif op == 'matmul_grad':
        c.update(x=(rng.integers(-4, 5, (2, n, 3)) / 2).astype(np.float32),
                 gout=(rng.integers(-4, 5, (2, m, 3)) / 2).astype(np.float32),
                 reduce=['sum', 'mean', 'min', 'max'][int(rng.integers(0, 4))])
    return c


def _tensor(ts, c, p, device):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    value = t(c[p + '_val']) if int(c[p + '_has']) else None
    return ts.SparseTensor(row=t(c[p + '_row']), col=t(c[p + '_col']), value=value,
                           sparse_sizes=(int(c[p + '_m']), int(c[p + '_n'])), is_sorted=True)


def _dump(out):
    if isinstance(out, torch.Tensor):
        return [out.detach().cpu().numpy()]
    row, col, value = out.coo()
    res = [row.cpu().numpy(), col.cpu().numpy(), np.array(out.sparse_sizes())]
    if value is not None:
        res.append(value.detach().cpu().numpy())
    return res


def run_case(ts, c, device):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    op = str(c['op'])
    A = _tensor(ts, c, 'A', device)
    if op == 'narrow':
        return _dump(A.narrow(int(c['dim']), int(c['start']), int(c['length'])))
    if op == 'index_select':
        return _dump(A.index_select(int(c['dim']), t(c['idx'])))
    if op == 'masked_select':
        return _dump(A.masked_select(int(c['dim']), t(c['mask'])))
    if op == 'remove_diag':
        return _dump(A.remove_diag(int(c['k'])))
    if op == 'set_diag':
        if int(c['A_has']) and c['A_val'].ndim > 1:  # the reference cannot broadcast its default ones here
            return _dump(A.fill_diag(1.0, int(c['k'])))
        return _dump(A.set_diag(None, int(c['k'])))
    if op == 'fill_diag':
        return _dump(A.fill_diag(0.75, int(c['k'])))
    if op == 'get_diag':
        return _dump(A.get_diag())
    if op == 'cat':
        B = _tensor(ts, c, 'B', device)
        dim = int(c['dim'])
        return _dump(ts.cat([A, B, A] if dim < 2 else [A, B], dim if dim < 2 else (0, 1)))
    if op == 'permute':
        return _dump(A.permute(t(c['perm'])))
    if op in ('mul_dense', 'add_dense'):
        vec = t(c['vec']).view(-1, 1) if int(c['rowwise']) else t(c['vec']).view(1, -1)
        if int(c['A_has']) and c['A_val'].ndim > 1:
            vec = vec.unsqueeze(-1)
        return _dump(ts.mul(A, vec) if op == 'mul_dense' else ts.add(A, vec))
    if op == 'add_sparse':
        return _dump(ts.add(A, _tensor(ts, c, 'B', device)))
    if op == 'mul_sparse':
        return _dump(ts.mul(A, _tensor(ts, c, 'B', device)))
    if op == 'reduce':
        return _dump(getattr(A, str(c['reduce']))(dim=int(c['dim'])))
    if op == 'transpose':
        return _dump(A.t())
    if op == 'coalesce':
        return _dump(A.coalesce(str(c['reduce'])))
    if op == 'masked_select_nnz':
        return _dump(A.masked_select_nnz(t(c['mask']), layout=str(c['layout'])))
    if op == 'index_select_nnz':
        return _dump(A.index_select_nnz(t(c['idx']), layout='coo'))
    if op == 'to_symmetric':
        return _dump(A.to_symmetric())
    if op == 'getitem':
        r0, r1 = sorted((int(c['r0']), int(c['r1'])))
        return _dump(A[r0:r1, t(c['mask'])])
    if op == 'saint':
        out, e = A.saint_subgraph(t(c['idx']))
        return _dump(out) + [e.cpu().numpy()]
    if op == 'sample_all':
        out, n_id = A.sample_adj(t(c['idx']), -1)
        return _dump(out) + [n_id.cpu().numpy()]
    if op == 'eye_matmul':
        return _dump(A.matmul(t(c['x']), reduce=str(c['reduce'])))
    if op in ('f_spmm', 'f_coalesce', 'f_transpose'):  # functional API on raw, shuffled (index, value)
        sh = t(c['shuffle'])
        index = torch.stack([t(c['A_row'])[sh], t(c['A_col'])[sh]])
        value = t(c['A_val'])[sh]
        m, n = int(c['A_m']), int(c['A_n'])
        if op == 'f_spmm':
            return [ts.spmm(index, value, m, n, t(c['x'])).cpu().numpy()]
        if op == 'f_coalesce':
            oi, ov = ts.coalesce(index, value, m, n, op=str(c['reduce']))
        else:
            oi, ov = ts.transpose(index, value, m, n)
        return [oi.cpu().numpy(), ov.cpu().numpy()]
    if op == 'f_spspmm':
        ia = torch.stack([t(c['A_row']), t(c['A_col'])])
        ib = torch.stack([t(c['B_row']), t(c['B_col'])])
        oi, ov = ts.spspmm(ia, t(c['A_val']), ib, t(c['B_val']), int(c['A_m']), int(c['A_n']), int(c['B_n']))
        return [oi.cpu().numpy(), ov.cpu().numpy()]
    if op == 'matmul_grad':
        value = t(c['A_val']).clone().requires_grad_()
        x = t(c['x']).clone().requires_grad_()
        A2 = ts.SparseTensor(row=t(c['A_row']), col=t(c['A_col']), value=value,
                             sparse_sizes=(int(c['A_m']), int(c['A_n'])), is_sorted=True)
        out = A2.matmul(x, reduce=str(c['reduce']))
        out.backward(t(c['gout']))
        gv = value.grad if value.grad is not None else torch.zeros_like(value)
        return [out.detach().cpu().numpy(), gv.cpu().numpy(), x.grad.cpu().numpy()]
    raise ValueError(op)

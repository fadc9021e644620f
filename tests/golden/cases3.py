"""Case list shared by the golden generator (run against the REFERENCE package) and by
tests/test_select_gpu.py (run against pytorch_sparse_amd): sub-matrix extraction, concatenation,
diagonal editing and element-wise ops -- SURVEY.md 8f ranks 2-3.

Every case is `fn(ts, I)`: `ts` is the package under test (the reference's `torch_sparse`, or
`pytorch_sparse_amd`), `I` the dict of inputs made by `make_inputs` / `load_inputs`.  A case returns a
SparseTensor or a dense Tensor.  Only the public API both packages share is used.
"""
import numpy as np
import torch


def _pattern(g, m, n, nnz):
    key = torch.randperm(m * n, generator=g)[:nnz].sort().values
    return key // n, key % n


def make_inputs():
    """Plain CPU tensors (saved once to py3_inputs.npz so both sides see identical bits)."""
    g = torch.Generator().manual_seed(1234)
    d = {}
    for name, (m, n, nnz) in dict(A=(37, 53, 400), B=(40, 40, 300), C=(20, 41, 150), D=(37, 10, 90),
                                  E=(37, 53, 350), Z=(6, 5, 0)).items():
        row, col = _pattern(g, m, n, nnz)
        d[name + '_row'], d[name + '_col'] = row, col
        d[name + '_val'] = torch.randint(-8, 9, (nnz, ), generator=g).to(torch.float32) / 4
        d[name + '_val2'] = torch.randint(-8, 9, (nnz, 3), generator=g).to(torch.float32) / 4
        d[name + '_size'] = torch.tensor([m, n])
    d['B_row'][:5] = d['B_col'][:5]  # B gets a few diagonal entries; re-sort / de-duplicate its keys
    key = (d['B_row'] * 40 + d['B_col']).unique()
    d['B_row'], d['B_col'] = key // 40, key % 40
    d['B_val'], d['B_val2'] = d['B_val'][:key.numel()], d['B_val2'][:key.numel()]
    d['idx_rows'] = torch.randint(0, 37, (50, ), generator=g)
    d['idx_cols'] = torch.randint(0, 53, (30, ), generator=g)
    d['mask_rows'] = torch.rand(37, generator=g) < 0.4
    d['mask_cols'] = torch.rand(53, generator=g) < 0.5
    d['idx_nnz'] = torch.randperm(400, generator=g)[:120].sort().values
    d['mask_nnz'] = torch.rand(400, generator=g) < 0.3
    d['perm_B'] = torch.randperm(40, generator=g)
    d['vec_rows'] = torch.randint(-4, 5, (37, 1), generator=g).to(torch.float32) / 2
    d['vec_cols'] = torch.randint(-4, 5, (1, 53), generator=g).to(torch.float32) / 2
    d['nnz_vec'] = torch.randint(-4, 5, (400, ), generator=g).to(torch.float32) / 2
    d['diag_vals'] = torch.randint(1, 9, (40, ), generator=g).to(torch.float32) / 2
    return d


def load_inputs(path):
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def tensors(ts, raw, device):
    """SparseTensors A..E,Z (1-D values), A2.. (2-D values), An.. (no values) + index / mask tensors."""
    I = {}
    for name in 'ABCDEZ':
        m, n = raw[name + '_size'].tolist()
        row, col = raw[name + '_row'].to(device), raw[name + '_col'].to(device)
        kw = dict(row=row, col=col, sparse_sizes=(m, n), is_sorted=True)
        I[name] = ts.SparseTensor(value=raw[name + '_val'].to(device), **kw)
        I[name + '2'] = ts.SparseTensor(value=raw[name + '_val2'].to(device), **kw)
        I[name + 'n'] = ts.SparseTensor(value=None, **kw)
    for k, v in raw.items():
        if k[1] != '_':
            I[k] = v.to(device)
    return I


def _slice_cases():
    c = []
    for v in ('A', 'A2', 'An'):
        c += [
            ('narrow0_%s' % v, lambda ts, I, v=v: ts.narrow(I[v], 0, 5, 20)),
            ('narrow0_all_%s' % v, lambda ts, I, v=v: I[v].narrow(0, 0, 37)),
            ('narrow0_neg_%s' % v, lambda ts, I, v=v: I[v].narrow(0, -7, 4)),
            ('narrow1_%s' % v, lambda ts, I, v=v: ts.narrow(I[v], 1, 7, 30)),
            ('narrow1_all_%s' % v, lambda ts, I, v=v: I[v].narrow(1, 0, 53)),
            ('narrow1_last_%s' % v, lambda ts, I, v=v: I[v].narrow(1, 52, 1)),
            ('select0_%s' % v, lambda ts, I, v=v: ts.select(I[v], 0, 3)),
            ('select1_%s' % v, lambda ts, I, v=v: I[v].select(1, 10)),
            ('index_select0_%s' % v, lambda ts, I, v=v: ts.index_select(I[v], 0, I['idx_rows'])),
            ('index_select1_%s' % v, lambda ts, I, v=v: I[v].index_select(1, I['idx_cols'])),
            ('masked_select0_%s' % v, lambda ts, I, v=v: ts.masked_select(I[v], 0, I['mask_rows'])),
            ('masked_select1_%s' % v, lambda ts, I, v=v: I[v].masked_select(1, I['mask_cols'])),
            ('index_select_nnz_coo_%s' % v,
             lambda ts, I, v=v: ts.index_select_nnz(I[v], I['idx_nnz'], layout='coo')),
            ('masked_select_nnz_coo_%s' % v,
             lambda ts, I, v=v: ts.masked_select_nnz(I[v], I['mask_nnz'], layout='coo')),
            ('masked_select_nnz_csc_%s' % v,
             lambda ts, I, v=v: I[v].masked_select_nnz(I['mask_nnz'], layout='csc')),
            ('getitem_slices_%s' % v, lambda ts, I, v=v: I[v][3:20, 5:30]),
            ('getitem_mask_idx_%s' % v, lambda ts, I, v=v: I[v][I['mask_rows'], I['idx_cols']]),
            ('getitem_int_%s' % v, lambda ts, I, v=v: I[v][4]),
            ('getitem_negslice_%s' % v, lambda ts, I, v=v: I[v][-10:, :-3]),
        ]
    c += [
        ('narrow_valuedim_A2', lambda ts, I: I['A2'].narrow(2, 1, 2)),
        ('index_select_valuedim_A2',
         lambda ts, I: I['A2'].index_select(2, torch.tensor([2, 0, 2], device=I['idx_rows'].device))),
        ('getitem_ellipsis_A2', lambda ts, I: I['A2'][..., 1:]),
        ('masked_select0_none', lambda ts, I: I['A'].masked_select(0, torch.zeros_like(I['mask_rows']))),
        ('masked_select1_all', lambda ts, I: I['A'].masked_select(1, torch.ones_like(I['mask_cols']))),
        ('index_select0_empty', lambda ts, I: I['A'].index_select(0, I['idx_rows'][:0])),
        ('index_select1_empty', lambda ts, I: I['A'].index_select(1, I['idx_cols'][:0])),
        ('narrow0_empty', lambda ts, I: I['A'].narrow(0, 10, 0)),
        ('narrow1_Z', lambda ts, I: I['Z'].narrow(1, 1, 3)),
        ('index_select0_Z', lambda ts, I: I['Z'].index_select(0, torch.tensor([5, 0, 0], device=I['idx_rows'].device))),
        ('permute_B', lambda ts, I: ts.permute(I['B'], I['perm_B'])),
        ('permute_Bn', lambda ts, I: I['Bn'].permute(I['perm_B'])),
    ]
    return c


def _cat_cases():
    c = []
    for v, s in (('', ''), ('2', '2'), ('n', 'n')):
        c += [
            ('cat0_%s' % (v or 'v'), lambda ts, I, s=s: ts.cat([I['A' + s], I['C' + s], I['E' + s]], 0)),
            ('cat1_%s' % (v or 'v'), lambda ts, I, s=s: ts.cat([I['A' + s], I['D' + s], I['C' + s]], 1)),
            ('catdiag_%s' % (v or 'v'), lambda ts, I, s=s: ts.cat([I['A' + s], I['B' + s], I['Z' + s]], (0, 1))),
        ]
    c += [
        ('cat1_single', lambda ts, I: ts.cat([I['A']], 1)),
        ('cat1_withZ', lambda ts, I: ts.cat([I['Z'], I['C'], I['Z']], 1)),
        ('cat_valuedim', lambda ts, I: ts.cat([I['A2'], I['A2']], 2)),
        ('cat0_then_narrow', lambda ts, I: ts.cat([I['A'], I['E']], 0).narrow(0, 37, 37)),
    ]
    return c


def _diag_cases():
    c = []
    for v in ('B', 'B2', 'Bn', 'A', 'An'):
        for k in (0, 2, -3):
            c += [
                ('remove_diag_%s_k%d' % (v, k), lambda ts, I, v=v, k=k: ts.remove_diag(I[v], k)),
                ('set_diag_%s_k%d' % (v, k), lambda ts, I, v=v, k=k: ts.set_diag(I[v], None, k)),
                ('fill_diag_%s_k%d' % (v, k), lambda ts, I, v=v, k=k: I[v].fill_diag(2.5, k)),
            ]
        c.append(('get_diag_%s' % v, lambda ts, I, v=v: ts.get_diag(I[v])))
        for k in (0, 2, -3):  # (|k| beyond the matrix crashes the reference) the reference's native op, same name in both packages
            c.append(('non_diag_mask_%s_k%d' % (v, k), lambda ts, I, v=v, k=k: torch.ops.torch_sparse.non_diag_mask(
                *ts.remove_diag(I[v], k).coo()[:2], I[v].sparse_size(0), I[v].sparse_size(1), k)))
    c += [
        ('set_diag_values_B', lambda ts, I: I['B'].set_diag(I['diag_vals'])),
        ('set_diag_values_B_k5', lambda ts, I: I['B'].set_diag(I['diag_vals'][:35], 5)),
        ('fill_diag_Z', lambda ts, I: I['Z'].fill_diag(1.0)),
        ('fill_diag_big_k', lambda ts, I: I['A'].fill_diag(1.0, 52)),
    ]
    return c


def _elementwise_cases():
    return [
        ('add_sparse', lambda ts, I: ts.add(I['A'], I['E'])),
        ('add_sparse_sizes', lambda ts, I: I['A'] + I['C']),
        ('add_sparse_noval', lambda ts, I: ts.add(I['An'], I['En'])),
        ('mul_sparse', lambda ts, I: ts.mul(I['A'], I['E'])),
        ('mul_sparse_sizes', lambda ts, I: I['A'] * I['C']),
        ('mul_rows', lambda ts, I: ts.mul(I['A'], I['vec_rows'])),
        ('mul_cols', lambda ts, I: I['A'] * I['vec_cols']),
        ('mul_cols_noval', lambda ts, I: I['An'] * I['vec_cols']),
        ('add_rows', lambda ts, I: ts.add(I['A'], I['vec_rows'])),
        ('add_cols_noval', lambda ts, I: I['An'] + I['vec_cols']),
        ('mul_nnz', lambda ts, I: ts.mul_nnz(I['A'], I['nnz_vec'], layout='coo')),
        ('add_nnz_csc', lambda ts, I: ts.add_nnz(I['A'], I['nnz_vec'], layout='csc')),
        ('mul_inplace', lambda ts, I: ts.mul_(I['A'].clone(), I['vec_rows'])),
        ('add_inplace', lambda ts, I: ts.add_(I['A'].clone(), I['vec_cols'])),
    ]


def all_cases():
    return _slice_cases() + _cat_cases() + _diag_cases() + _elementwise_cases()

"""Fuzz of the record-writing min / max forward (tsamd_spmm_minmax_records) against the id route: random row-length laws
(uniform, power-law, a few hub rows, many empty rows), sizes, widths, dtypes, values, batches -- records word for word
against tsamd_spmm_minmax_winrec on the ids, outputs and pull gradients bit for bit.  Not collected by pytest:
    python tests/fuzz_records.py [seconds] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from pytorch_sparse_amd import _native as nat  # noqa: E402
from tests.test_records_gpu import _csc, _winrec_no_value  # noqa: E402

DEV = 'cuda'
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    n = int(torch.randint(1, 6000, (1, ), generator=g))
    ncol = int(torch.randint(1, 6000, (1, ), generator=g))
    law = int(torch.randint(0, 4, (1, ), generator=g))
    if law == 0:
        deg = torch.randint(0, 60, (n, ), generator=g)
    elif law == 1:
        deg = (torch.rand(n, generator=g) ** -1.2).clamp(max=20000).long() - 1
    elif law == 2:
        deg = torch.randint(0, 12, (n, ), generator=g)
        hubs = torch.randint(0, n, (3, ), generator=g)
        deg[hubs] = torch.randint(200, 9000, (3, ), generator=g)
    else:
        deg = torch.randint(0, 300, (n, ), generator=g) * (torch.rand(n, generator=g) < 0.1).long()
    rp = torch.zeros(n + 1, dtype=torch.int64)
    rp[1:] = deg.cumsum(0)
    E = int(rp[-1])
    if E == 0 or E > 3_000_000:
        continue
    c = torch.randint(0, ncol, (E, ), generator=g)
    K = int([36, 64, 100, 128, 128, 128, 160, 256][int(torch.randint(0, 8, (1, ), generator=g))])
    dtype = [torch.bfloat16, torch.float16, torch.float32][int(torch.randint(0, 3, (1, ), generator=g))]
    has_value = bool(torch.randint(0, 2, (1, ), generator=g))
    batch = (2, ) if (E < 200_000 and bool(torch.randint(0, 4, (1, ), generator=g) == 0)) else ()
    reduce = 'max' if bool(torch.randint(0, 2, (1, ), generator=g)) else 'min'
    x = (torch.randn(*batch, ncol, K, generator=g) * 2).round().to(dtype)  # many ties
    if bool(torch.randint(0, 5, (1, ), generator=g) == 0):
        x[..., torch.randint(0, ncol, (max(1, ncol // 50), ), generator=g), :] = float('nan')
    v = (torch.randn(E, generator=g)).to(dtype) if has_value else None
    colptr, perm, row = _csc(rp, c, ncol)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    out_a, arg = nat.spmm_minmax_arg32(d(rp), d(c), d(v), d(x), reduce)
    out_r, rec = nat.spmm_minmax_records(d(rp), d(c), d(v), d(x), reduce, d(row), zero=True)
    want = nat.spmm_minmax_winrec(d(row), d(v), arg, K) if has_value else _winrec_no_value(d(row), arg, K, dtype)
    tag = dict(n=n, ncol=ncol, law=law, E=E, K=K, dtype=str(dtype), has_value=has_value, batch=batch, reduce=reduce, seed=seed, case=cases)
    assert torch.equal(out_a.view(torch.uint8), out_r.view(torch.uint8)), tag
    assert torch.equal(rec, want), (tag, int((rec != want).sum()))
    gr = torch.randn(*batch, n, K, generator=g).to(dtype).to(DEV)
    wv = has_value and (K * x.element_size()) % 16 == 0
    gv_a, gm_a = nat.spmm_minmax_bw_csc(d(rp), d(c), d(v), d(x), gr, arg, d(colptr), d(perm), d(row), want_value=wv, want_mat=True)
    gv_r, gm_r = nat.spmm_minmax_bw_csc_records(d(rp), d(c), has_value, d(x), gr, rec, d(colptr), d(perm), d(row), want_value=wv)
    assert torch.equal(gm_a.view(torch.uint8), gm_r.view(torch.uint8)), tag
    if wv:
        assert torch.equal(gv_a.view(torch.uint8), gv_r.view(torch.uint8)), tag
    cases += 1
print('fuzz_records: %d cases in %.0f s, seed %d: all bit-identical' % (cases, time.time() - t0, seed))

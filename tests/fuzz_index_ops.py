"""One-off differential fuzzer for the index pipelines at medium sizes: SparseTensor construction
(sort), coalesce (all reductions), transpose / csr2csc, SpSpMM (all three row-size classes), against
the numpy oracle (itself pinned by the reference-generated fixtures).  Small-integer values make every
sum exact, so everything is compared bit for bit.  Not collected by pytest; lives under tests/ because
it drives the oracle.  Usage (GPU box):  python tests/fuzz_index_ops.py [--cases 120] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402

DEV = 'cuda'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def draw_coo(rng, m, n, nnz, law):
    if law == 0:    # uniform, few duplicates
        row, col = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
    elif law == 1:  # heavy duplication
        row, col = rng.integers(0, max(m // 50, 1), nnz), rng.integers(0, max(n // 50, 1), nnz)
    elif law == 2:  # hub rows and hub columns
        row = np.minimum(rng.zipf(1.6, nnz) - 1, m - 1)
        col = np.minimum(rng.zipf(1.4, nnz) - 1, n - 1)
    else:           # already sorted, no duplicates
        key = np.unique(rng.integers(0, m * n, nnz))
        row, col = key // n, key % n
    return row.astype(np.int64), col.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=120)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    fails = 0
    for case in range(args.cases):
        m = int(rng.choice([1, 7, 300, 20_000, 150_000]))
        n = int(rng.choice([1, 5, 400, 30_000, 100_000]))
        nnz = int(rng.choice([0, 1, 50, 5_000, 300_000, 1_500_000]))
        law = int(rng.integers(0, 4))
        row, col = draw_coo(rng, m, n, nnz, law)
        nnz = row.size
        D = int(rng.choice([0, 1, 1, 2]))
        val = None if D == 0 else rng.integers(-8, 9, (nnz, ) if D == 1 else (nnz, D)).astype(np.float32)
        tag = 'case %d: m=%d n=%d nnz=%d law=%d D=%d' % (case, m, n, nnz, law, D)
        try:
            A = ts.SparseTensor(row=dev(row), col=dev(col), value=None if val is None else dev(val), sparse_sizes=(m, n))
            r, c, perm = npo.sort_coo(row, col, m, n)
            gr, gc, gv = A.coo()
            assert np.array_equal(gr.cpu().numpy(), r) and np.array_equal(gc.cpu().numpy(), c), 'sort'
            if val is not None:
                assert np.array_equal(gv.cpu().numpy(), val[perm]), 'sort values (stable)'
            assert np.array_equal(A.storage.rowptr().cpu().numpy(),
                                  np.concatenate([[0], np.cumsum(np.bincount(r, minlength=m))])), 'rowptr'
            for op in ('add', 'mean', 'min', 'max'):
                if val is None and op != 'add':
                    continue
                er, ec, ev = npo.coalesce(row, col, val, m, n, op)
                C = A.coalesce('sum' if op == 'add' else op)
                cr, cc, cv = C.coo()
                assert np.array_equal(cr.cpu().numpy(), er) and np.array_equal(cc.cpu().numpy(), ec), 'coalesce ' + op
                if val is not None:
                    if op == 'mean':
                        assert np.allclose(cv.cpu().numpy(), ev, rtol=1e-6, atol=1e-6), 'coalesce mean'
                    else:
                        assert np.array_equal(cv.cpu().numpy(), ev), 'coalesce ' + op
            Cs = A.coalesce('sum')
            T = Cs.t()
            tr, tc, tv = T.coo()
            cr, cc, cv = (x.cpu().numpy() if x is not None else None for x in Cs.coo())
            o = np.lexsort((cr, cc))
            assert np.array_equal(tr.cpu().numpy(), cc[o]) and np.array_equal(tc.cpu().numpy(), cr[o]), 'transpose'
            if cv is not None:
                assert np.array_equal(tv.cpu().numpy(), cv[o]), 'transpose values'
            assert np.array_equal(Cs.storage.csr2csc().cpu().numpy(), o), 'csr2csc'
            # SpSpMM: Cs (m x n) times a random (n x k) matrix; exact with small integers
            if D <= 1 and Cs.nnz() <= 400_000:
                k = int(rng.choice([1, 50, 20_000]))
                nb = int(rng.choice([0, 100, 200_000]))
                rb, cb = draw_coo(rng, n, k, nb, int(rng.integers(0, 4)))
                vb = rng.integers(-4, 5, rb.size).astype(np.float64)
                B = ts.SparseTensor(row=dev(rb), col=dev(cb), value=dev(vb), sparse_sizes=(n, k)).coalesce('sum')
                br, bc, bv = (x.cpu().numpy() for x in B.coo())
                Ad = Cs if cv is not None else Cs.fill_value(1.0)
                Ad = Ad.set_value(Ad.storage.value().double(), layout='coo')
                av = Ad.storage.value().cpu().numpy()
                cnt = np.bincount(br, minlength=n)[cc] if cc.size else np.zeros(0, np.int64)
                if int(cnt.sum()) <= 30_000_000:
                    er, ec, ev = npo.spspmm(cr, cc, av, br, bc, bv, m, n, k)
                    P = Ad @ B
                    pr, pc, pv = P.coo()
                    assert np.array_equal(pr.cpu().numpy(), er) and np.array_equal(pc.cpu().numpy(), ec), 'spspmm index'
                    assert np.array_equal(pv.cpu().numpy(), ev), 'spspmm values'
        except Exception as exc:  # noqa: BLE001
            fails += 1
            print('FAIL', tag, '::', type(exc).__name__, str(exc)[:300], flush=True)
        if case % 20 == 19:
            print('... %d cases, %d failures' % (case + 1, fails), flush=True)
    print('fuzz_index_ops: %d cases, %d failures (seed %d)' % (args.cases, fails, args.seed), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()

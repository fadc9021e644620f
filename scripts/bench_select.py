"""Sub-matrix extraction / concatenation / diagonal timings on one MI355X.

Workload: the config-2 graph (R-MAT scale 20, edge factor 20: 20.0 M entries, fp32 values).  Every line carries
  ms        wall time of the public API call (includes its one host sync),
  ref_ms    the same call written the way the reference writes it (ATen boolean-mask / fancy-index
            compositions, torch_sparse/{index_select,masked_select,narrow,cat,diag}.py) on the SAME
            GPU -- the comparison a user switching packages would see,
  gbs       algorithmic bytes (read the entries that are looked at once, write the output once) / ms.
Prints one JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')


def wall(fn, iters=7, warm=2):
    for _ in range(warm):
        fn()
    best = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best.append(time.perf_counter() - t0)
    best.sort()
    return best[len(best) // 2] * 1e3


scale = int(os.environ.get('SCALE', 20))
rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
n = 1 << scale
E = c.numel()
val = synth.values(E, device=dev)
A = ts.SparseTensor(rowptr=rp, col=c, value=val, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
row = A.storage.row()
rowcount = A.storage.rowcount()
g = torch.Generator(device='cpu').manual_seed(1)
idx = torch.randint(0, n, (n // 2, ), generator=g).to(dev)
mask = (torch.rand(n, generator=g) < 0.5).to(dev)


def emit(name, ms, ref_ms, bytes_alg, **kw):
    print(json.dumps(dict(bench=name, E=E, ms=round(ms, 3), ref_ms=None if ref_ms is None else round(ref_ms, 3),
                          speedup=None if ref_ms is None else round(ref_ms / ms, 2),
                          gbs=round(bytes_alg / ms / 1e6, 1), **kw)), flush=True)


# ---- index_select(0): half as many rows as the matrix has, with repeats
def ref_index_select0():
    cnt = rowcount[idx]
    ptr = c.new_zeros(idx.numel() + 1)
    torch.cumsum(cnt, 0, out=ptr[1:])
    r = torch.arange(idx.numel(), device=dev).repeat_interleave(cnt)
    perm = torch.arange(r.numel(), device=dev)
    perm += (rp[idx] - ptr[:-1])[r]  # gather_csr
    return r, c[perm], val[perm]


out = A.index_select(0, idx)
T = out.nnz()
emit('index_select_rows', wall(lambda: A.index_select(0, idx)), wall(ref_index_select0),
     idx.numel() * 24 + T * (8 + 4) + T * (8 + 8 + 8 + 4), picked=idx.numel(), out_nnz=T)


# ---- masked_select(0) / (1)
def ref_masked_select0():
    cnt = rowcount[mask]
    m = mask[row]
    r = torch.arange(cnt.numel(), device=dev).repeat_interleave(cnt)
    return r, c[m], val[m]


def ref_masked_select1():  # the reference goes through the CSC view and sorts back (masked_select.py:39-63);
    m = mask[c]            # give it the cheaper order-preserving formulation instead
    newcol = torch.cumsum(mask, 0) - 1
    return row[m], newcol[c[m]], val[m]


T = A.masked_select(0, mask).nnz()
emit('masked_select_rows', wall(lambda: A.masked_select(0, mask)), wall(ref_masked_select0),
     E * (16 + 1) + n * 9 + T * (8 + 8 + 4 + 4), out_nnz=T)
T = A.masked_select(1, mask).nnz()
emit('masked_select_cols', wall(lambda: A.masked_select(1, mask)), wall(ref_masked_select1),
     E * (16 + 1) + n * 9 + T * (8 + 8 + 4 + 4), out_nnz=T)


# ---- narrow(1): the middle half of the columns
def ref_narrow1():
    m = (c >= n // 4) & (c < n // 4 + n // 2)
    return row[m], c[m] - n // 4, val[m]


T = A.narrow(1, n // 4, n // 2).nnz()
emit('narrow_cols', wall(lambda: A.narrow(1, n // 4, n // 2)), wall(ref_narrow1),
     E * 16 + T * (8 + 8 + 4 + 4), out_nnz=T)

# ---- narrow(0): views
emit('narrow_rows', wall(lambda: A.narrow(0, n // 4, n // 2)), wall(lambda: (rp[n // 4:n // 4 + n // 2 + 1] - rp[n // 4], int(rp[n // 4]), int(rp[n // 4 + n // 2]))),
     (n // 2) * 16)

# ---- cat(1) of two column blocks vs concatenate + sort (cat.py:160-165 -> storage.py:149-162)
L, R = A.narrow(1, 0, n // 2), A.narrow(1, n // 2, n - n // 2)
for t in (L, R):
    t.storage.rowptr(), t.storage.row()


def ref_cat1():
    r = torch.cat([L.storage.row(), R.storage.row()])
    cc = torch.cat([L.storage.col(), R.storage.col() + n // 2])
    v = torch.cat([L.storage.value(), R.storage.value()])
    perm = (r * n + cc).argsort()
    return r[perm], cc[perm], v[perm]


emit('cat_cols', wall(lambda: ts.cat([L, R], 1)), wall(ref_cat1), E * (16 + 4) + E * (8 + 8 + 8 + 4) + n * 48)
parts = [A.narrow(0, i * (n // 8), n // 8) for i in range(8)]
emit('cat_rows_8_shards', wall(lambda: ts.cat(parts, 0)),
     wall(lambda: (torch.cat([p.storage.col() for p in parts]), torch.cat([p.storage.value() for p in parts]),
                   torch.cat([p.storage.rowptr()[1:] for p in parts]))), E * 24)


# ---- fill_diag (GCN self loops): remove + merge
def ref_fill_diag():
    inv = row != c
    r, cc, v = row[inv], c[inv], val[inv]
    # the reference's native mask op only exists on the device as part of its CUDA build; emulate
    # its result with the position formula and do the four boolean scatters it is followed by
    slot = torch.arange(r.numel(), device=dev) + r + (r < cc).to(torch.long)
    m = torch.zeros(r.numel() + n, dtype=torch.bool, device=dev)
    m[slot] = True
    nr = r.new_empty(m.numel()); nr[m] = r; nr[~m] = torch.arange(n, device=dev)
    nc = r.new_empty(m.numel()); nc[m] = cc; nc[~m] = torch.arange(n, device=dev)
    nv = v.new_empty(m.numel()); nv[m] = v; nv[~m] = 1.0
    return nr, nc, nv


T = A.fill_diag(1.0).nnz()
emit('fill_diag', wall(lambda: A.fill_diag(1.0)), wall(ref_fill_diag), E * 16 + E * 28 * 2 + T * 28, out_nnz=T)

# ---- sparse + sparse, sparse * sparse (A with its transpose)
At = A.t()
Ac = A.coalesce()
Atc = At.coalesce()
emit('add_sparse', wall(lambda: ts.add(A, At), iters=3), wall(lambda: torch.sparse_coo_tensor(
    torch.stack([torch.cat([row, At.storage.row()]), torch.cat([c, At.storage.col()])]),
    torch.cat([val, At.storage.value()]), (n, n)).coalesce(), iters=3), 2 * E * 20 * 2)
emit('mul_sparse', wall(lambda: ts.mul(Ac, Atc), iters=3), None, 2 * Ac.nnz() * 20 * 2)

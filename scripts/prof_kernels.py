"""Profile target for the PMC passes (scripts/profile_pmc.sh): every kernel family of the hot path, a few
launches each, in one process.  Workloads are separated in the dispatch stream by a MARKER launch
(tsamd gather_rows of one row -- no workload uses that kernel) before and after, so scripts/summarize_pmc.py can attribute
the per-dispatch counter rows of `rocprofv3 --pmc ... --kernel-include-regex tsamd` to a workload by order.

Prints one JSON line: the workload order with the byte counts each kernel is priced against (algorithmic,
SURVEY 8d) and, where it is known exactly, the PHYSICAL byte count of a streaming kernel (the calibration
points for FETCH_SIZE / WRITE_SIZE: spmm_permute_rows_kernel reads and writes B*N*K*s bytes exactly once).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import _native as nat, synth  # noqa: E402

dev = torch.device('cuda:0')
which = set(sys.argv[1:])
reps = 2
plan = []
_one = torch.zeros(1, 4, device=dev)
_idx = torch.zeros(1, dtype=torch.long, device=dev)


def marker():
    torch.ops.tsamd.gather_rows(_one, _idx)


def run(label, fn, **info):
    """warm call, MARKER, `reps` calls, MARKER: the dispatches between the two markers are the workload's;
    whatever runs between workloads (set-up, warm calls) falls into a bucket that is thrown away."""
    if which and label not in which:
        return
    fn()  # warm (allocator, caches)
    torch.cuda.synchronize()
    marker()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    marker()
    plan.append(dict(label=label, launches=reps, **info))
    plan.append(dict(label='_discard', launches=1))


def balg(E, M, K, s, has_value, minmax):
    return E * (8 + (s if has_value else 0) + K * s) + (M + 1) * 8 + M * K * s + (M * K * 8 if minmax else 0)


# ---- north star + uniform control (the calibration graph: no hub reuse) -------------------------------------
n = 1 << 21
rp, c = synth.rmat_csr(21, 20, seed=0, device=dev)
E = c.numel()
v = synth.values(E, device=dev)
x = synth.features(n, 128, device=dev)
run('ns_sum_f32_F128', lambda: nat.spmm(rp, c, v, x, 'sum'), edges=E, algorithmic_bytes=balg(E, n, 128, 4, True, False),
    b_min=E * 12 + (n + 1) * 8 + 2 * n * 128 * 4,
    calibration=dict(kernel='spmm_permute_rows_kernel', read_bytes=n * 128 * 4, write_bytes=n * 128 * 4))
rpu, cu = synth.uniform_degree_csr(n, n, 20, seed=5, device=dev)
vu = synth.values(cu.numel(), seed=6, device=dev)
run('control_uniform_sum_f32_F128', lambda: nat.spmm(rpu, cu, vu, x, 'sum'), edges=cu.numel(),
    algorithmic_bytes=balg(cu.numel(), n, 128, 4, True, False), b_min=cu.numel() * 12 + (n + 1) * 8 + 2 * n * 128 * 4,
    note='uniform columns: 42 M gathers over 2 M rows of 512 B = 20 visits per row spread over the whole launch; '
         'X (1.07 GB) is 4x the Infinity Cache, so nearly every gather is a fabric read: fetch ~ algorithmic bytes')
del rp, c, v, rpu, cu, vu, x
torch.cuda.empty_cache()

# ---- config 2 / 3 graph ----------------------------------------------------------------------------------------
n = 1 << 20
rp, c = synth.rmat_csr(20, 20, seed=0, device=dev)
E = c.numel()
row = nat.ptr2ind(rp, E)
x = synth.features(n, 64, device=dev)
g = synth.features(n, 64, seed=3, device=dev)
run('c2_value_bw_f32_F64', lambda: nat.spmm_value_bw(row, rp, c, x, g, 'sum'), edges=E,
    algorithmic_bytes=E * (16 + 64 * 4 + 4) + n * 64 * 4)
if not which or 'c2_train_step' in which:
    _v = synth.values(E, device=dev).requires_grad_()
    _A = ts.SparseTensor(rowptr=rp, col=c, value=_v, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    _A.storage.fill_cache_()
    _xr = x.clone().requires_grad_()

    def _step():
        _xr.grad = None
        _v.grad = None
        _A.matmul(_xr, 'sum').backward(g)
    _step()
    _step()
    torch.ops.tsamd.operand_cache(False)
    run('c2_train_step', _step, edges=E, algorithmic_bytes=3 * balg(E, n, 64, 4, True, False))
    torch.ops.tsamd.operand_cache(False)
xb = synth.features(n, 128, dtype=torch.bfloat16, device=dev)
gb = synth.features(n, 128, seed=3, dtype=torch.bfloat16, device=dev)
run('c3_max_fw_bf16_F128', lambda: nat.spmm(rp, c, None, xb, 'max'), edges=E, algorithmic_bytes=balg(E, n, 128, 2, False, True))
_, arg = nat.spmm(rp, c, None, xb, 'max')
run('c3_max_bw_atomic_bf16_F128', lambda: nat.spmm_minmax_bw(rp, c, None, xb, gb, arg, want_value=False, want_mat=True),
    edges=E, algorithmic_bytes=n * 128 * (8 + 2 * 2) + 2 * n * 128 * 2)
if hasattr(nat, 'spmm_minmax_bw_csc'):
    A = ts.SparseTensor(rowptr=rp, col=c, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    colptr, perm = A.storage.colptr(), A.storage.csr2csc()
    run('c3_max_bw_pull_bf16_F128', lambda: nat.spmm_minmax_bw_csc(rp, c, None, xb, gb, arg, colptr, perm, row,
                                                                    want_value=False, want_mat=True),
        edges=E, algorithmic_bytes=n * 128 * (8 + 2 * 2) + 2 * n * 128 * 2)
    if hasattr(nat, 'spmm_minmax_records'):
        # round 6: the forward that leaves the backward's winner records, and the backward that starts on them
        run('c3_max_fw_records_bf16_F128', lambda: nat.spmm_minmax_records(rp, c, None, xb, 'max', row), edges=E,
            algorithmic_bytes=balg(E, n, 128, 2, False, True))
        _, _rec = nat.spmm_minmax_records(rp, c, None, xb, 'max', row)
        run('c3_max_bw_records_bf16_F128', lambda: nat.spmm_minmax_bw_csc_records(rp, c, False, xb, gb, _rec, colptr, perm, row),
            edges=E, algorithmic_bytes=n * 128 * (8 + 2 * 2) + 2 * n * 128 * 2)
        del _rec
    vb = synth.values(E, dtype=torch.bfloat16, device=dev)
    _, argv = nat.spmm(rp, c, vb, xb, 'max')
    run('c3_max_bw_pull_val_bf16_F128', lambda: nat.spmm_minmax_bw_csc(rp, c, vb, xb, gb, argv, colptr, perm, row,
                                                                        want_value=True, want_mat=True),
        edges=E, algorithmic_bytes=n * 128 * (8 + 2 * 2) + 2 * n * 128 * 2)
    run('c3_value_bw_plain_bf16_F128', lambda: nat.spmm_value_bw(row, rp, c, xb, gb, 'sum'), edges=E,
        algorithmic_bytes=E * (16 + 128 * 2 + 2) + n * 128 * 2)
    if not which or 'c3_matmul_step_val_bf16_F128' in which:
        # round 5: the drop-in training step (int32 winner ids inside the node, pull with grad_value)
        _vm = vb.clone().requires_grad_()
        _Am = ts.SparseTensor(rowptr=rp, col=c, value=_vm, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
        _Am.storage.colptr(), _Am.storage.csr2csc(), _Am.storage.row()
        _xm = xb.clone().requires_grad_()

        def _mstep():
            _xm.grad = None
            _vm.grad = None
            _Am.matmul(_xm, 'max').backward(gb)
        _mstep()
        run('c3_matmul_step_val_bf16_F128', _mstep, edges=E,
            algorithmic_bytes=balg(E, n, 128, 2, True, True) + n * 128 * (8 + 2 * 2) + 2 * n * 128 * 2)
        del _Am, _vm, _xm
    del A, colptr, perm, vb, argv
del rp, c, row, x, g, xb, gb, arg
torch.cuda.empty_cache()

# ---- sort / coalesce and SpSpMM on the config-4 input ------------------------------------------------------------
m = 500000
r4, c4 = synth.uniform_edges(m, m, 7500000, seed=0, device=dev)
v4 = synth.values(7500000, device=dev)
run('sort_coo_7m5', lambda: torch.ops.tsamd.sort_coo(r4, c4, m, m, True), entries=7500000,
    algorithmic_bytes=7500000 * 16 * 2 + 7500000 * 8)
def _ctor():
    B_ = ts.SparseTensor(row=r4, col=c4, value=v4, sparse_sizes=(m, m))
    B_.storage.rowptr()


_index4 = torch.stack([r4, c4])
run('coalesce_7m5', lambda: ts.coalesce(_index4, v4, m, m), entries=7500000, algorithmic_bytes=7500000 * 20 * 2)
run('construct_7m5', _ctor, entries=7500000, algorithmic_bytes=7500000 * 20 * 2 + (m + 1) * 8)
run('transpose_7m5', lambda: ts.transpose(_index4, v4, m, m), entries=7500000, algorithmic_bytes=7500000 * 20 * 2)
if not which or 'sort_coo_75m' in which:
    _m75 = 1 << 22
    _r75, _c75 = synth.uniform_edges(_m75, _m75, 75000000, seed=0, device=dev)
    run('sort_coo_75m', lambda: torch.ops.tsamd.sort_coo(_r75, _c75, _m75, _m75, True), entries=75000000,
        algorithmic_bytes=75000000 * 16 * 2 + 75000000 * 8)
    del _r75, _c75
    torch.cuda.empty_cache()
A = ts.SparseTensor(row=r4, col=c4, value=v4, sparse_sizes=(m, m)).coalesce()
At = A.t()
rpB = At.storage.rowptr()
P = int((rpB[A.storage.col() + 1] - rpB[A.storage.col()]).sum())
C = A @ At
run('c4_spspmm', lambda: A @ At, products=P, algorithmic_bytes=A.nnz() * 12 + P * 12 + C.nnz() * 20)
del A, At, C, r4, c4, v4
torch.cuda.empty_cache()
if 'uniform40_spspmm' in which:
    mu = 200000
    ru, cu_ = synth.uniform_edges(mu, mu, 40 * mu, seed=0, device=dev)
    Au = ts.SparseTensor(row=ru, col=cu_, value=synth.values(ru.numel(), device=dev), sparse_sizes=(mu, mu)).coalesce()
    Atu = Au.t()
    Pu = int((Atu.storage.rowptr()[Au.storage.col() + 1] - Atu.storage.rowptr()[Au.storage.col()]).sum())
    nnzCu = (Au @ Atu).nnz()
    run('uniform40_spspmm', lambda: Au @ Atu, products=Pu, algorithmic_bytes=Au.nnz() * 12 + Pu * 12 + nnzCu * 20)
    del Au, Atu
if not which or 'stress_spspmm' in which:
    rp, c = synth.rmat_csr(19, 8, seed=0, device=dev)
    A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(c.numel(), device=dev), sparse_sizes=(1 << 19, 1 << 19),
                        is_sorted=True, trust_data=True)
    At = A.t()
    rpB = At.storage.rowptr()
    P = int((rpB[A.storage.col() + 1] - rpB[A.storage.col()]).sum())
    nnzC = (A @ At).nnz()
    run('stress_spspmm', lambda: A @ At, products=P, algorithmic_bytes=A.nnz() * 12 + P * 12 + nnzC * 20)
print(json.dumps(dict(marker_kernel='gather_rows_kernel', workloads=plan)))

"""BASELINE.json configs C2 / C3 / C4 (and the north-star parity leg) at their stated size: one
function per config that runs the HIP path, times it with HIP events, times the CPU baseline
beside it and compares the WHOLE output with the checker.  Shared by ``bench.py`` (the
``secondary`` array of the JSON line) and ``tests/test_configs_gpu.py``.

TEST / BENCH INFRASTRUCTURE: this is the only module besides the tests that drives ``oracle/``
(the compiled reference ``oracle/_ref`` when it was built, else the C restatement); the product
never imports it.  The reference benchmark does the same thing in the same run
(/root/reference/benchmark/main.py:36-58: time, then compare with an independent
implementation); the tolerances are the reference test-suite's (test/test_matmul.py:45-51) or
tighter and are written next to each check.
"""
import os
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


# ------------------------------------------------------------------------------------------------
def b_alg(E, M, K, esize, has_value, minmax):
    """Algorithmic bytes of one SpMM (no-reuse gather model)."""
    return E * (8 + (esize if has_value else 0) + K * esize) + (M + 1) * 8 + M * K * esize + \
        (M * K * 8 if minmax else 0)


def b_min(E, M, N, K, esize, has_value, minmax):
    """Compulsory bytes of one SpMM: every operand read once, the result written once."""
    return E * (8 + (esize if has_value else 0)) + (M + 1) * 8 + N * K * esize + M * K * esize + \
        (M * K * 8 if minmax else 0)


def gpu_ms(fn, iters=10, warm=2):
    """Median HIP-event time of fn() on the current stream (the stream the kernels launch on)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def _ref_ops():
    try:
        from oracle import ref
        if ref.available():
            return ref.ops()
    except Exception:
        pass
    return None


def cpu_time(fn, budget_s=25.0, max_reps=3):
    """Best wall time of fn() on the host: one run, then up to max_reps more while they fit the
    budget.  Returns (seconds, last result, runs)."""
    t0 = time.perf_counter()
    res = fn()
    first = best = time.perf_counter() - t0
    reps = 0
    while reps < max_reps and (reps + 2) * first < budget_s:
        t0 = time.perf_counter()
        res = fn()
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return best, res, reps + 1


def ref_spmm_cpu(rp, c, v, x, reduce):
    """The reference's CPU kernel (compiled from /root/reference by oracle/build_ref.py) on host
    tensors, or the C restatement when oracle/_ref is absent -> (out, arg or None, kind)."""
    r = _ref_ops()
    if r is not None:
        if reduce == 'sum':
            return r.spmm_sum(None, rp, c, v, None, None, x), None, 'reference'
        if reduce == 'mean':
            return r.spmm_mean(None, rp, c, v, None, None, None, x), None, 'reference'
        o, a = (r.spmm_min if reduce == 'min' else r.spmm_max)(rp, c, v, x)
        return o, a, 'reference'
    from tests.util import oracle_spmm
    o, a = oracle_spmm(rp, c, v, x, reduce)
    return o, a, 'port'


# ------------------------------------------------------------------------------------------------
# parity statistics
# ------------------------------------------------------------------------------------------------
def sum_parity(out_gpu, rp, c, v, x, ref_out=None, tol=1e-5):
    """fp32 SpMM-sum/mean, WHOLE output against the reference CPU kernel's fp32 output:
         max_err_over_l1 = max |a - b| / sum_e |v_e x_e|      (bar: 1e-5, tests/util.py)
         max_rel / frac_rel_gt_tol: the element-wise |a - b| / |b| statistic (it blows up on
             cancelled sums, for the reference's own fp32 result as well -- which is why the same
             figures are given for both against the fp64 result of the reference kernel).
    All operands are host tensors except out_gpu."""
    a = out_gpu.detach().cpu()
    if ref_out is None:
        ref_out = ref_spmm_cpu(rp, c, v, x, 'sum')[0]
    l1 = ref_spmm_cpu(rp, c, None if v is None else v.abs(), x.abs(), 'sum')[0].double()
    exact = ref_spmm_cpu(rp, c, None if v is None else v.double(), x.double(), 'sum')[0]
    l1c = l1.clamp(min=1e-30)
    d = (a.double() - ref_out.double()).abs()
    relb = d / ref_out.double().abs().clamp(min=1e-30)
    e_ours = (a.double() - exact).abs() / l1c
    e_ref = (ref_out.double() - exact).abs() / l1c
    res = dict(elements=int(a.numel()), against='reference CPU kernel, whole output',
               max_err_over_l1=float((d / l1c).max()), tol_over_l1=tol,
               max_rel_vs_ref=float(relb.max()), frac_rel_vs_ref_gt_1e_5=float((relb > 1e-5).double().mean()),
               ours_vs_fp64_over_l1=float(e_ours.max()), ref_vs_fp64_over_l1=float(e_ref.max()))
    res['ok'] = bool(res['max_err_over_l1'] <= tol and res['ours_vs_fp64_over_l1'] <= tol)
    return res


def rmat_graph(scale, ef, dev, seed=0):
    from pytorch_sparse_amd import synth
    rp, c = synth.rmat_csr(scale, ef, seed=seed, device=dev)
    return rp, c, 1 << scale


# ------------------------------------------------------------------------------------------------
# C2: CSR SpMM-sum 1M x 1M R-MAT ~20 nnz/row, F = 64 fp32
# ------------------------------------------------------------------------------------------------
def run_c2(dev, cpu=True, iters=20):
    from pytorch_sparse_amd import synth
    rp, c, n = rmat_graph(20, 20, dev)
    E, K = c.numel(), 64
    v = synth.values(E, device=dev)
    x = synth.features(n, K, device=dev)
    op = torch.ops.torch_sparse.spmm_sum
    ms = gpu_ms(lambda: op(None, rp, c, v, None, None, x), iters=iters)
    out = op(None, rp, c, v, None, None, x)
    ba = b_alg(E, n, K, 4, True, False)
    res = dict(config='c2', workload='configs[1]: CSR SpMM-sum 2^20 x 2^20 R-MAT (E=%d), F=64 fp32' % E,
               dtype='f32', ms=round(ms, 4), gedges_per_s=round(E / ms / 1e6, 3),
               roofline=dict(bound='hbm', algorithmic_bytes=ba, b_min=b_min(E, n, n, K, 4, True, False),
                             achieved=round(ba / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                             frac=round(ba / ms / 1e6 / HBM_PEAK_GBS, 4), scope='whole op (all kernels of the call)'))
    if cpu:
        rpc, cc, vc, xc = rp.cpu(), c.cpu(), v.cpu(), x.cpu()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        t, (ro, _, kind), runs = cpu_time(lambda: ref_spmm_cpu(rpc, cc, vc, xc, 'sum'))
        res['cpu_baseline'] = dict(value=round(E / t / 1e9, 4), unit='GEdges/s', cores=cores, kind=kind,
                                   ms=round(t * 1e3, 2), sample='full workload, best of %d' % runs)
        res['parity'] = sum_parity(out, rpc, cc, vc, xc, ro)
    return res


# ------------------------------------------------------------------------------------------------
# C3: CSR SpMM-max + backward (arg path), same graph, F = 128 bf16
# ------------------------------------------------------------------------------------------------
def minmax_bw_bound(col, value, grad_out, arg, n, dtype):
    """Exact fp64 grad_mat of the reference's formulas (csrc/spmm.cpp:224-239) on the GPU by ATen
    scatter_add_ (an independent implementation), with the error bound of accumulating
    `cnt` addends in the element type in any order: (cnt + 1) * u * sum|terms|."""
    E = col.numel()
    invalid = arg == E
    a = arg.masked_fill(invalid, 0)
    ind = col[a]
    w = value[a].double() if value is not None else 1.0
    term = (w * grad_out.double()).masked_fill(invalid, 0)
    K = grad_out.size(-1)
    z = lambda: torch.zeros(n, K, dtype=torch.float64, device=arg.device)  # noqa: E731
    exact = z().scatter_add_(0, ind, term)
    l1 = z().scatter_add_(0, ind, term.abs())
    cnt = z().scatter_add_(0, ind, (~invalid).double())
    u = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -11,
         torch.bfloat16: 2.0 ** -8}[dtype]
    floor = 2.0 ** -25 if dtype == torch.float16 else 1e-40  # half a subnormal step per addition
    return exact, (cnt + 1) * (u * l1 * 1.01 + floor)


def run_c3(dev, has_value, cpu=True, iters=10):
    from pytorch_sparse_amd import synth
    from pytorch_sparse_amd import _native as nat
    dtype = torch.bfloat16
    rp, c, n = rmat_graph(20, 20, dev)
    E, K = c.numel(), 128
    v = synth.values(E, dtype=dtype, device=dev) if has_value else None
    x = synth.features(n, K, dtype=dtype, device=dev)
    g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
    op = torch.ops.torch_sparse.spmm_max
    fw_ms = gpu_ms(lambda: op(rp, c, v, x), iters=iters)
    out, arg = op(rp, c, v, x)
    bw = lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=has_value, want_mat=True)  # noqa: E731
    bw_ms = gpu_ms(bw, iters=iters)
    gval, gmat = bw()
    # autograd wiring of the drop-in op: out.backward(g) must give the same grad_mat kernel result
    xr = x.clone().requires_grad_()
    vr = v.clone().requires_grad_() if has_value else None
    o2, _ = op(rp, c, vr, xr)
    o2.backward(g)
    ba_fw = b_alg(E, n, K, 2, has_value, True)
    ba_bw = n * K * (8 + 2 * 2) + 2 * n * K * 2  # M*F*(8+2s) read + scatter RMW 2*M*F*s
    res = dict(config='c3', has_value=has_value,
               workload='configs[2]: CSR SpMM-max + backward, 2^20 R-MAT (E=%d), F=128 bf16, %s' % (
                   E, 'with values' if has_value else 'value-less'),
               dtype='bf16', fw_ms=round(fw_ms, 4), bw_ms=round(bw_ms, 4), gedges_per_s_fw=round(E / fw_ms / 1e6, 3),
               roofline=dict(bound='hbm', algorithmic_bytes=ba_fw, achieved=round(ba_fw / fw_ms / 1e6, 1),
                             peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ba_fw / fw_ms / 1e6 / HBM_PEAK_GBS, 4),
                             scope='forward, whole op'),
               roofline_bw=dict(bound='hbm atomics', algorithmic_bytes=ba_bw, achieved=round(ba_bw / bw_ms / 1e6, 1),
                                peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ba_bw / bw_ms / 1e6 / HBM_PEAK_GBS, 4),
                                note='bounded by the device-scope atomic rate (20 G 64-byte segments/s, '
                                     'profiles/r02_ubench_atomics.csv), not by bytes'))
    par = dict()
    # backward: grad_mat against the fp64 formulas within the rounding bound of its arithmetic
    exact, bound = minmax_bw_bound(c, v, g, arg, n, dtype)
    err = (gmat.double() - exact).abs()
    par['grad_mat_max_err_over_bound'] = float((err / bound).max())
    par['grad_mat_autograd_equal_bound'] = float(((xr.grad.double() - exact).abs() / bound).max())
    del exact, bound, err
    if has_value:
        invalid = arg == E
        a = arg.masked_fill(invalid, 0)
        term = (x.gather(0, c[a]).double() * g.double()).masked_fill(invalid, 0)
        ev = torch.zeros(E, dtype=torch.float64, device=dev).scatter_add_(0, a.flatten(), term.flatten())
        l1 = torch.zeros(E, dtype=torch.float64, device=dev).scatter_add_(0, a.flatten(), term.abs().flatten())
        # fp32 accumulation in LDS, one rounding to bf16
        bnd = 2.0 ** -8 * ev.abs() * 1.01 + 1e-5 * l1 + 1e-30
        par['grad_value_max_err_over_bound'] = float(((gval.double() - ev).abs() / bnd).max())
        par['grad_value_autograd_max_err_over_bound'] = float(((vr.grad.double() - ev).abs() / bnd).max())
        del term, ev, l1, bnd
    if cpu:
        rpc, cc, xc = rp.cpu(), c.cpu(), x.cpu()
        vc = None if v is None else v.cpu()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        t, (ro, ra, kind), runs = cpu_time(lambda: ref_spmm_cpu(rpc, cc, vc, xc, 'max'), budget_s=20.0, max_reps=1)
        res['cpu_baseline'] = dict(value=round(E / t / 1e9, 4), unit='GEdges/s (forward)', cores=cores, kind=kind,
                                   ms=round(t * 1e3, 2), sample='full workload forward, best of %d' % runs)
        # forward: bit-exact values and arg_out over all M*K elements
        par['elements'] = int(ra.numel())
        par['arg_out_mismatches'] = int((arg.cpu() != ra).sum())
        par['out_bit_mismatches'] = int((out.cpu().view(torch.int16) != ro.view(torch.int16)).sum())
        par['against'] = 'reference CPU kernel (forward, bit-exact); fp64 formulas of csrc/spmm.cpp:204-242 (backward)'
    par['ok'] = bool(par.get('arg_out_mismatches', 0) == 0 and par.get('out_bit_mismatches', 0) == 0 and
                     par['grad_mat_max_err_over_bound'] <= 1.0 and par['grad_mat_autograd_equal_bound'] <= 1.0 and
                     par.get('grad_value_max_err_over_bound', 0.0) <= 1.0 and
                     par.get('grad_value_autograd_max_err_over_bound', 0.0) <= 1.0)
    res['parity'] = par
    return res


# ------------------------------------------------------------------------------------------------
# C4: SpSpMM A * A^T, 500k x 500k, ~15 nnz/row (uniform)
# ------------------------------------------------------------------------------------------------
def spspmm_inputs(dev, kind='c4'):
    import pytorch_sparse_amd as ts
    from pytorch_sparse_amd import synth
    if kind == 'c4':
        m = n = 500000
        row, col = synth.uniform_edges(m, n, 7500000, seed=0, device=dev)
        A = ts.SparseTensor(row=row, col=col, value=synth.values(row.numel(), device=dev), sparse_sizes=(m, n)).coalesce()
    else:  # stress row of SURVEY 8d: R-MAT scale 19 (hub rows: products explode quadratically)
        rp, c = synth.rmat_csr(19, 8, seed=0, device=dev)
        m = n = 1 << 19
        A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(c.numel(), device=dev), sparse_sizes=(m, n),
                            is_sorted=True, trust_data=True)
    return A, A.t()


def spspmm_properties(A, B, C, sample_rows=48):
    """Size-independent checks of C = A * B where the host SpGEMM would take minutes (the R-MAT stress
    product): structure (rows sorted by column, no duplicates), the two checksums
        sum_j C_ij = sum_k A_ik (sum_j B_kj)      and      sum_ij C_ij = sum_k (sum_i A_ik)(sum_j B_kj)
    evaluated with ATen in fp64 (independent of the kernels under test), and `sample_rows` random rows
    (plus the longest one) compared exactly with torch.sparse.mm on the host (the reference's call)."""
    rowA, colA, valA = A.coo()
    rowB, colB, valB = B.coo()
    row, col, val = C.coo()
    m = A.sparse_size(0)
    dev = row.device
    same_row = row[1:] == row[:-1]
    sorted_unique = bool(((col[1:] > col[:-1]) | ~same_row).all())
    b_rowsum = torch.zeros(B.sparse_size(0), dtype=torch.float64, device=dev).index_add_(0, rowB, valB.double())
    want_rows = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, rowA, valA.double() * b_rowsum[colA])
    l1_rows = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, rowA, valA.double().abs() * torch.zeros_like(
        b_rowsum).index_add_(0, rowB, valB.double().abs())[colA])
    got_rows = torch.zeros(m, dtype=torch.float64, device=dev).index_add_(0, row, val.double())
    row_err = float(((got_rows - want_rows).abs() / l1_rows.clamp(min=1e-30)).max())
    total_err = float(abs(got_rows.sum() - want_rows.sum()) / l1_rows.sum())
    # sampled rows, exactly
    rp = C.storage.rowptr()
    deg = rp[1:] - rp[:-1]
    g = torch.Generator().manual_seed(0)
    pick = torch.cat([torch.randint(0, m, (sample_rows, ), generator=g), deg.argmax().cpu().view(1)]).unique()
    Asub = A.index_select(0, pick.to(dev))
    ref = torch.sparse.mm(Asub.cpu().to_torch_sparse_coo_tensor(), B.cpu().to_torch_sparse_coo_tensor()).coalesce()
    Csub = C.index_select(0, pick.to(dev))
    r2, c2, v2 = Csub.coo()
    idx_ok = torch.equal(torch.stack([r2, c2]).cpu(), ref._indices())
    l1s = (Asub.set_value(Asub.storage.value().abs(), 'coo') @ B.set_value(valB.abs(), 'coo')).storage.value()
    verr = float(((v2.cpu().double() - ref._values().double()).abs() / l1s.cpu().double().clamp(min=1e-30)).max()) \
        if idx_ok else float('inf')
    ok = sorted_unique and row_err <= 1e-5 and total_err <= 1e-6 and idx_ok and verr <= 1e-5
    return dict(against='structure + fp64 ATen checksums (all rows) + torch.sparse.mm on %d sampled rows' % pick.numel(),
                nnz=int(col.numel()), rows_sorted_unique=sorted_unique, row_sum_max_err_over_l1=row_err,
                total_sum_rel_err=total_err, sampled_rows_index_bit_exact=bool(idx_ok),
                sampled_rows_value_max_err_over_l1=verr, ok=bool(ok))


def run_spspmm(dev, kind='c4', cpu=True, iters=5):
    A, At = spspmm_inputs(dev, kind)
    rpB = At.storage.rowptr()
    colA = A.storage.col()
    P = int((rpB[colA + 1] - rpB[colA]).sum())
    ms = gpu_ms(lambda: A @ At, iters=iters, warm=1)
    C = A @ At
    nnzA, nnzC = A.nnz(), C.nnz()
    comp = nnzA * (8 + 4) + P * (8 + 4) + nnzC * (16 + 4)  # SURVEY 8d compulsory bytes
    name = ('configs[3]: SpSpMM A*A^T, 500k x 500k uniform' if kind == 'c4' else
            'stress: SpSpMM A*A^T, R-MAT scale 19 edge factor 8')
    res = dict(config=kind, workload='%s (nnzA=%d, products=%d, nnzC=%d), fp32' % (name, nnzA, P, nnzC),
               dtype='f32', ms=round(ms, 4), gproducts_per_s=round(P / ms / 1e6, 3),
               roofline=dict(bound='hbm', algorithmic_bytes=comp, achieved=round(comp / ms / 1e6, 1),
                             peak=HBM_PEAK_GBS, unit='GB/s', frac=round(comp / ms / 1e6 / HBM_PEAK_GBS, 4),
                             scope='whole op incl. its host syncs; bytes = compulsory (SURVEY 8d)'))
    if not cpu and kind != 'c4':
        res['parity'] = spspmm_properties(A, At, C)
    if cpu:
        Ac, Bc = A.cpu().to_torch_sparse_coo_tensor(), At.cpu().to_torch_sparse_coo_tensor()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        Cc = torch.sparse.mm(Ac, Bc)  # what the reference calls (torch_sparse/matmul.py:104)
        t = time.perf_counter() - t0
        res['cpu_baseline'] = dict(value=round(P / t / 1e9, 4), unit='GProducts/s', cores=cores, kind='reference',
                                   ms=round(t * 1e3, 1), sample='full workload, torch.sparse.mm on the host, 1 run')
        Cc = Cc.coalesce()
        row, col, val = C.coo()
        idx_ok = torch.equal(torch.stack([row, col]).cpu(), Cc._indices())
        ev = Cc._values().double()
        # fp32 sums of ~1 product each, different order: 1e-5 relative to the L1 mass (>= |value|)
        l1 = (A.set_value(A.storage.value().abs(), 'coo') @ At.set_value(At.storage.value().abs(), 'coo')).storage.value()
        err = float(((val.cpu().double() - ev).abs() / l1.cpu().double().clamp(min=1e-30)).max()) if idx_ok else float('inf')
        res['parity'] = dict(against='torch.sparse.mm (CPU), whole output', nnz=int(nnzC), index_bit_exact=bool(idx_ok),
                             value_max_err_over_l1=err, tol_over_l1=1e-5, ok=bool(idx_ok and err <= 1e-5))
    return res

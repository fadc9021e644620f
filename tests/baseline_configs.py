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


def gpu_ms_stream(fn, iters=20, warm=2):
    """Mean HIP-event time per call of `iters` calls queued back to back (how the headline times its steps): the
    host runs ahead, so the launch latency in front of a call's first kernel -- which gpu_ms() includes, every
    one of its calls starts on an empty queue -- overlaps the previous call."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


_PMC = None


def pmc_traffic(workload, kernel_substr):
    """Counter-measured fabric bytes per call of one kernel (FETCH_SIZE x 2 + WRITE_SIZE, calibrated in
    profiles/traffic_ns.json) from the committed builder-run profiles -- the newest of profiles/r06_pmc.json / r05_pmc.json /
    r04_pmc.json / r03_pmc.json that holds the workload AND a kernel of that name -- NOT measured in this run (PMC passes need
    rocprofv3).  kernel_substr '*' = every kernel of the workload summed (a sort is build + passes).
    -> dict for a roofline's `traffic` fields, or {} when no file has it."""
    global _PMC
    if _PMC is None:
        import json
        _PMC = []
        for name in ('r06_pmc.json', 'r05_pmc.json', 'r04_pmc.json', 'r03_pmc.json'):
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', name)
            try:
                _PMC.append((name, json.load(open(path))))
            except Exception:
                pass
    for name, prof in _PMC:
        w = prof.get('workloads', {}).get(workload)
        if not w:
            continue
        src = 'profiles/%s (%s), rocprofv3 --pmc builder run, NOT measured in this run' % (name, workload)
        if kernel_substr == '*':
            ks = [v for v in w['kernels'].values() if 'fabric_bytes_per_call_x2_rule' in v]
            if ks:
                us = sum(v['us_per_call'] for v in ks)
                nbytes = sum(v['fabric_bytes_per_call_x2_rule'] for v in ks)
                return dict(traffic=nbytes, traffic_kernel='all %d kernels of the call' % len(ks), traffic_kernel_us=round(us, 1),
                            traffic_tb_per_s=round(nbytes / us / 1e6, 3) if us > 0 else None, traffic_source=src)
            continue
        for k, v in w['kernels'].items():
            if kernel_substr in k and 'fabric_bytes_per_call_x2_rule' in v:
                return dict(traffic=v['fabric_bytes_per_call_x2_rule'], traffic_kernel=k, traffic_kernel_us=round(v['us_per_call'], 1),
                            traffic_tb_per_s=v.get('fabric_tb_per_s_x2_rule'), traffic_source=src)
    return {}


class operand_cache(object):
    """with operand_cache(False): every product copies its dense operand again (tsamd_spmm), so that a timing loop
    over the same X measures ALL the work of a call -- the ops' default since round 4 (the cache is opt-in);
    with operand_cache(True): the opt-in behaviour.  The previous setting is restored on exit."""

    def __init__(self, enabled):
        self.enabled = enabled
        self.was = False

    def __enter__(self):
        self.was = bool(torch.ops.tsamd.operand_cache(self.enabled)[0])

    def __exit__(self, *exc):
        torch.ops.tsamd.operand_cache(self.was)


def _ref_ops():
    try:
        from oracle import ref
        if ref.available():
            return ref.ops()
    except Exception:
        pass
    return None


def host_threads(n=None):
    """All cores for a timed CPU-baseline leg (n = None: os.cpu_count()), a bounded pool (32) for everything else:
    ATen's host ops on the statistics-sized tensors of these rows run many times slower with 256 threads than with 32."""
    torch.set_num_threads(n or (os.cpu_count() or 1))


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
    host_threads(32)  # (the caller raised the count for this leg: back to the bounded pool)
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
def sum_parity(out_gpu, rp, c, v, x, ref_out=None, tol=1e-5, fp64_leg=True):
    """fp32 SpMM-sum/mean, WHOLE output against the reference CPU kernel's fp32 output.

    The north star's "fp32 within 1e-5 rel" is checked in the two readings it admits:
      * norm-wise (the pass / fail bar, tests/util.py):  |a - b| <= 1e-5 * sum_e |v_e x_e|  for EVERY element,
        for ours-vs-reference and for ours-vs-fp64;
      * element-wise |a - b| <= 1e-5 * |b|, which no fp32 kernel with a different summation order than
        spmm_cpu.cpp:73-87 can meet on cancelled sums (the reference's own result misses it against fp64 just
        as often) -- so it is required only where the sum is well conditioned:
            n_rel_gt_1e_5_where_ref_ge_1e_1_l1   elements with |b| >= 0.1 L1 violating it   (must be 0)
            n_rel_gt_1e_5_where_ref_ge_1e_2_l1   the same for |b| >= 0.01 L1               (reported; the worst
                                                 case there is 100 * (2.0e-7 + 3.7e-7) = 5.7e-5)
        and the unconditional figures max_rel_vs_ref / frac_rel_vs_ref_gt_1e_5 are printed next to the same
        two figures of the reference against fp64.
    All operands are host tensors except out_gpu.  fp64_leg=False (the configs[4]-share TEST: 537 M elements; the
    bench line's row of the same workload keeps the full statistics) drops the third host product -- the fp64 one --
    and what is derived from it; the whole output is still compared with the reference's fp32 output."""
    dev = out_gpu.device  # the statistics run with ATen in fp64 ON THE DEVICE (independent of the kernels under test):
    a = out_gpu.detach()  # 268-537 M elements take seconds there instead of a minute on the host
    if ref_out is None:
        ref_out = ref_spmm_cpu(rp, c, v, x, 'sum')[0]
    l1 = ref_spmm_cpu(rp, c, None if v is None else v.abs(), x.abs(), 'sum')[0].to(dev).double()
    refd = ref_out.to(dev).double()
    if not fp64_leg:
        l1c = l1.clamp(min=1e-30)
        d = (a.double() - refd).abs()
        bad = d > 1e-5 * refd.abs()
        res = dict(elements=int(a.numel()),
                   against='reference CPU kernel (csrc/cpu/spmm_cpu.cpp via oracle/_ref), whole output; criterion: '
                           '|gpu - ref| <= 1e-5 * sum_e|v_e x_e| for every element, plus |gpu - ref| <= 1e-5 * |ref| for '
                           'every element with |ref| >= 0.1 * sum_e|v_e x_e| (fp64 statistics: headline row)',
                   max_err_over_l1=float((d / l1c).max()), tol_over_l1=tol,
                   n_rel_gt_1e_5_where_ref_ge_1e_1_l1=int((bad & (refd.abs() >= 1e-1 * l1)).sum()))
        res['ok'] = bool(res['max_err_over_l1'] <= tol and res['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0)
        return res
    exact = ref_spmm_cpu(rp, c, None if v is None else v.double(), x.double(), 'sum')[0].to(dev)
    l1c = l1.clamp(min=1e-30)
    ad = a.double()
    d = (ad - refd).abs()
    relb = d / refd.abs().clamp(min=1e-30)
    bad = relb > 1e-5
    e_ours = (ad - exact).abs() / l1c
    e_ref = (refd - exact).abs() / l1c
    rel_ref64 = (refd - exact).abs() / exact.abs().clamp(min=1e-30)
    res = dict(elements=int(a.numel()),
               against='reference CPU kernel (csrc/cpu/spmm_cpu.cpp via oracle/_ref), whole output; criterion: '
                       '|gpu - ref| <= 1e-5 * sum_e|v_e x_e| for every element (and the same against fp64), '
                       'plus |gpu - ref| <= 1e-5 * |ref| for every element with |ref| >= 0.1 * sum_e|v_e x_e|',
               max_err_over_l1=float((d / l1c).max()), tol_over_l1=tol,
               max_rel_vs_ref=float(relb.max()), frac_rel_vs_ref_gt_1e_5=float(bad.double().mean()),
               n_rel_gt_1e_5_where_ref_ge_1e_1_l1=int((bad & (refd.abs() >= 1e-1 * l1)).sum()),
               n_rel_gt_1e_5_where_ref_ge_1e_2_l1=int((bad & (refd.abs() >= 1e-2 * l1)).sum()),
               n_elements_ref_ge_1e_2_l1=int((refd.abs() >= 1e-2 * l1).sum()),
               ref_vs_fp64_frac_rel_gt_1e_5=float((rel_ref64 > 1e-5).double().mean()),
               ours_vs_fp64_over_l1=float(e_ours.max()), ref_vs_fp64_over_l1=float(e_ref.max()))
    del d, relb, bad, e_ours, e_ref, rel_ref64
    # SURVEY 8(d)'s bound as written: |a - b| <= 1e-5 * max(|b|, 1e-5 * ||row||_1).  It is element-wise relative
    # down to 1e-5 of the row's L1 mass, i.e. it also bites on cancelled sums -- where ANY fp32 summation order,
    # the reference's own sequential one included, misses it against the exact (fp64) sum.  So the counts are
    # reported for the three pairs, and the pass / fail statement is "ours violates it against fp64 no more often
    # than the reference's own fp32 kernel does".
    def literal(a_, b_):
        return int(((a_ - b_).abs() > 1e-5 * torch.maximum(b_.abs(), 1e-5 * l1)).sum())
    res['survey_8d_literal_bound'] = dict(
        bound='|a - b| <= 1e-5 * max(|b|, 1e-5 * L1(row))',
        n_viol_ours_vs_ref=literal(ad, refd), n_viol_ours_vs_fp64=literal(ad, exact),
        n_viol_ref_vs_fp64=literal(refd, exact))
    lb = res['survey_8d_literal_bound']
    lb['ours_le_ref'] = bool(lb['n_viol_ours_vs_fp64'] <= lb['n_viol_ref_vs_fp64'])
    res['ok'] = bool(res['max_err_over_l1'] <= tol and res['ours_vs_fp64_over_l1'] <= tol and
                     res['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0 and lb['ours_le_ref'])
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
    with operand_cache(False):  # all the work of a call in every timed iteration
        ms = gpu_ms(lambda: op(None, rp, c, v, None, None, x), iters=iters)
        ms_b2b = gpu_ms_stream(lambda: op(None, rp, c, v, None, None, x), iters=iters)
    out = op(None, rp, c, v, None, None, x)
    with operand_cache(True):  # opt-in operand cache: same X again
        ms_rep = gpu_ms(lambda: op(None, rp, c, v, None, None, x), iters=iters)
    ba = b_alg(E, n, K, 4, True, False)
    res = dict(config='c2', workload='configs[1]: CSR SpMM-sum 2^20 x 2^20 R-MAT (E=%d), F=64 fp32' % E,
               dtype='f32', ms=round(ms, 4), gedges_per_s=round(E / ms / 1e6, 3), ms_repeated_operand=round(ms_rep, 4),
               ms_back_to_back=round(ms_b2b, 4),
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
# C5 share: what ONE of the 8 GPUs of configs[4] multiplies -- 2^21 rows, ~32 nnz/row, F = 256 fp32 (1 KB rows: the
# 1024-item branch of plan_partition) -- plus the row-sharded path with P = 8 logical ranks on this one device
# ------------------------------------------------------------------------------------------------
def run_c5_share(dev, cpu=True, iters=10, logical_ranks=8, fp64_leg=True):
    from pytorch_sparse_amd import synth
    from pytorch_sparse_amd import _native as nat
    from pytorch_sparse_amd.parallel import narrow_rows, partition_rows
    rp, c, n = rmat_graph(21, 32, dev)
    E, K = c.numel(), 256
    v = synth.values(E, device=dev)
    x = synth.features(n, K, device=dev)
    op = torch.ops.torch_sparse.spmm_sum
    ms = gpu_ms(lambda: op(None, rp, c, v, None, None, x), iters=iters)
    ms_b2b = gpu_ms_stream(lambda: op(None, rp, c, v, None, None, x), iters=iters)
    out = op(None, rp, c, v, None, None, x)
    ba = b_alg(E, n, K, 4, True, False)
    res = dict(config='c5_share',
               workload='configs[4] per-GPU share: CSR SpMM-sum 2^21 x 2^21 R-MAT edge factor 32 (E=%d), F=256 fp32' % E,
               dtype='f32', ms=round(ms, 4), ms_back_to_back=round(ms_b2b, 4), gedges_per_s=round(E / ms / 1e6, 3),
               roofline=dict(bound='hbm', algorithmic_bytes=ba, b_min=b_min(E, n, n, K, 4, True, False),
                             achieved=round(ba / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                             frac=round(ba / ms / 1e6 / HBM_PEAK_GBS, 4), scope='whole op (all kernels of the call)'))
    # P logical ranks, one after the other on this device (SURVEY 8e validation): every rank multiplies its
    # nnz-balanced row block (narrow semantics, global column ids) with the full X
    ranges = partition_rows(rp, logical_ranks, 'nnz')
    l1 = nat.spmm(rp, c, v.abs(), x.abs(), 'sum')[0]
    worst, t_blocks = 0.0, []
    mx_full, ma_full = nat.spmm(rp, c, v, x, 'max')
    max_equal = True
    for s_, e_ in ranges:
        rpl, cl, vl = narrow_rows(rp, c, v, s_, e_)
        rpl = rpl.contiguous()
        blk = nat.spmm(rpl, cl, vl, x, 'sum')[0]
        t_blocks.append(gpu_ms(lambda: nat.spmm(rpl, cl, vl, x, 'sum'), iters=3, warm=1))
        worst = max(worst, float(((blk.double() - out[s_:e_].double()).abs() / l1[s_:e_].double().clamp(min=1e-30)).max()))
        bm, bam = nat.spmm(rpl, cl, vl, x, 'max')
        e0 = int(rp[s_])
        bam = torch.where(bam == cl.numel(), torch.full_like(bam, E), bam + e0)  # block-local entry ids -> global
        max_equal = max_equal and bool(torch.equal(bm, mx_full[s_:e_])) and bool(torch.equal(bam, ma_full[s_:e_]))
        del blk, bm, bam
    del mx_full, ma_full, l1
    res['row_sharded'] = dict(logical_ranks=logical_ranks, block_ms=[round(t, 4) for t in t_blocks],
                              sum_of_blocks_ms=round(sum(t_blocks), 4), max_block_ms=round(max(t_blocks), 4),
                              sum_max_err_over_l1_vs_unsharded=worst, max_and_arg_bit_identical_to_unsharded=max_equal,
                              note='sum: a long row is cut by the merge-path partition at other places inside a block, so its '
                                   'fp32 partial sums associate differently (<= 1e-5 of the L1 mass required); max: exact')
    par = dict()
    if cpu:
        rpc, cc, vc, xc = rp.cpu(), c.cpu(), v.cpu(), x.cpu()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        t, (ro, _, kind), runs = cpu_time(lambda: ref_spmm_cpu(rpc, cc, vc, xc, 'sum'), budget_s=20.0, max_reps=1)
        res['cpu_baseline'] = dict(value=round(E / t / 1e9, 4), unit='GEdges/s', cores=cores, kind=kind,
                                   ms=round(t * 1e3, 2), sample='full workload, best of %d' % runs)
        par = sum_parity(out, rpc, cc, vc, xc, ro, fp64_leg=fp64_leg)
    par['row_sharded_ok'] = bool(worst <= 1e-5 and max_equal)
    par['ok'] = bool(par.get('ok', True) and par['row_sharded_ok'])
    res['parity'] = par
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
    with operand_cache(False):
        fw_ms = gpu_ms(lambda: op(rp, c, v, x), iters=iters)
        fw_b2b = gpu_ms_stream(lambda: op(rp, c, v, x), iters=iters)
    out, arg = op(rp, c, v, x)
    # backward, both routes: the pull over the cached CSC arrays (what adj.matmul(x, 'max').backward() runs,
    # tsamd_spmm_minmax_bw_csc) and the scatter with packed atomics behind the bare 4-argument op
    import pytorch_sparse_amd as ts
    vr = v.clone().requires_grad_() if has_value else None
    A = ts.SparseTensor(rowptr=rp, col=c, value=vr, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    st = A.storage
    colptr, perm, row = st.colptr(), st.csr2csc(), st.row()
    bw = lambda: nat.spmm_minmax_bw_csc(rp, c, v, x, g, arg, colptr, perm, row, want_value=has_value, want_mat=True)  # noqa: E731
    bw_ms = gpu_ms(bw, iters=iters)
    gval, gmat = bw()
    # round 6: the backward as the matmul node runs it -- on the winner records its forward left (tsamd_spmm_minmax_records)
    out_rec, rec = nat.spmm_minmax_records(rp, c, v, x, 'max', row)
    bw_rec = lambda: nat.spmm_minmax_bw_csc_records(rp, c, has_value, x, g, rec, colptr, perm, row, want_value=has_value)  # noqa: E731
    bw_rec_ms = gpu_ms(bw_rec, iters=iters)
    with operand_cache(False):
        fw_rec_ms = gpu_ms_stream(lambda: nat.spmm_minmax_records(rp, c, v, x, 'max', row), iters=iters)
    gval_r, gmat_r = bw_rec()
    records_same = bool(torch.equal(out_rec.view(torch.int16), out.view(torch.int16)) and
                        torch.equal(gmat_r.view(torch.int16), gmat.view(torch.int16)) and
                        (not has_value or torch.equal(gval_r.view(torch.int16), gval.view(torch.int16))))
    del rec, out_rec, gval_r, gmat_r
    bw_atomic = lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=has_value, want_mat=True)  # noqa: E731
    bw_atomic_ms = gpu_ms(bw_atomic, iters=iters)
    gval_a, gmat_a = bw_atomic()
    deterministic = bool(torch.equal(bw()[1].view(torch.int16), gmat.view(torch.int16)))
    # autograd wiring of the drop-in front-end: adj.matmul(x, 'max').backward(g) takes the pull route
    xr = x.clone().requires_grad_()
    # (128 two-byte features: the front-end takes the pull with and without grad_value; asking for deterministic
    # algorithms -- BEFORE the forward, when the CSC arrays are handed over -- pins it whatever the rule says)
    torch.use_deterministic_algorithms(has_value)
    try:
        o2 = A.matmul(xr, 'max')
        o2.backward(g)
    finally:
        torch.use_deterministic_algorithms(False)
    same_as_op = bool(torch.equal(xr.grad.view(torch.int16), gmat.view(torch.int16)))
    # the drop-in training step, as a caller of the reference writes it: adj.matmul(x, 'max') and .backward(g).  Since
    # round 5 the node keeps its winners as int32 ids (tsamd_spmm_minmax_arg32) and takes the pull with grad_value too
    xt = x.clone().requires_grad_()

    def train_fw():
        return A.matmul(xt, 'max')

    def train_step():
        xt.grad = None
        if vr is not None:
            vr.grad = None
        A.matmul(xt, 'max').backward(g)

    with operand_cache(False):
        train_fw_ms = gpu_ms(train_fw, iters=iters)
        train_step_ms = gpu_ms(train_step, iters=iters)
    train_step()
    train_same = bool(torch.equal(xt.grad.view(torch.int16), gmat.view(torch.int16)))
    if vr is not None:
        train_same = train_same and bool(torch.equal(vr.grad.view(torch.int16), gval.view(torch.int16)))
    # ... and the bare reference op (no CSC arrays) the scatter route
    xr2 = x.clone().requires_grad_()
    o3, _ = op(rp, c, v, xr2)
    o3.backward(g)
    ba_fw = b_alg(E, n, K, 2, has_value, True)
    ba_bw = n * K * (8 + 2 * 2) + 2 * n * K * 2  # M*F*(8+2s) read + scatter RMW 2*M*F*s
    res = dict(config='c3', has_value=has_value,
               workload='configs[2]: CSR SpMM-max + backward, 2^20 R-MAT (E=%d), F=128 bf16, %s' % (
                   E, 'with values' if has_value else 'value-less'),
               dtype='bf16', fw_ms=round(fw_ms, 4), fw_ms_back_to_back=round(fw_b2b, 4), bw_ms=round(bw_ms, 4), gedges_per_s_fw=round(E / fw_ms / 1e6, 3),
               matmul_fw_ms=round(train_fw_ms, 4), matmul_fw_bw_ms=round(train_step_ms, 4),
               bw_records_ms=round(bw_rec_ms, 4), fw_records_ms_back_to_back=round(fw_rec_ms, 4),
               roofline_bw_records=dict(bound='hbm', algorithmic_bytes=ba_bw, achieved=round(ba_bw / bw_rec_ms / 1e6, 1),
                                        peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ba_bw / bw_rec_ms / 1e6 / HBM_PEAK_GBS, 4),
                                        scope='the pull backward inside SparseTensor.matmul (round 6): the forward left the '
                                              'winner records (fw_records_ms_back_to_back against fw_ms_back_to_back), the '
                                              'backward starts at the masked sum'),
               roofline=dict(bound='hbm', algorithmic_bytes=ba_fw, achieved=round(ba_fw / fw_ms / 1e6, 1),
                             peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ba_fw / fw_ms / 1e6 / HBM_PEAK_GBS, 4),
                             scope='forward, whole op', **pmc_traffic('c3_max_fw_bf16_F128', 'spmm_merge_kernel')),
               bw_atomic_ms=round(bw_atomic_ms, 4),
               roofline_bw=dict(bound='hbm', algorithmic_bytes=ba_bw, achieved=round(ba_bw / bw_ms / 1e6, 1),
                                peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ba_bw / bw_ms / 1e6 / HBM_PEAK_GBS, 4),
                                scope='pull route (winner masks + masked merge-path SpMM over the cached CSC arrays); '
                                      'bytes = SURVEY 8d\'s scatter model M F (8 + 2 s) + 2 M F s, which the pull does '
                                      'not follow: it gathers one grad_out row per entry, E (16 + F s + F / 8)',
                                bytes_pull_model=int(E * (16 + K * 2 + K // 8) + n * K * 8 + n * K * 2),
                                frac_pull_model=round((E * (16 + K * 2 + K // 8) + n * K * 8 + n * K * 2) / bw_ms / 1e6 / HBM_PEAK_GBS, 4),
                                atomic_route_frac=round(ba_bw / bw_atomic_ms / 1e6 / HBM_PEAK_GBS, 4),
                                traffic_pull=dict(masked_merge=pmc_traffic('c3_max_bw_pull_bf16_F128', 'spmm_merge_kernel'),
                                                  winner_records=pmc_traffic('c3_max_bw_pull_bf16_F128', 'minmax_winrec_kernel')),
                                traffic_atomic=pmc_traffic('c3_max_bw_atomic_bf16_F128', 'spmm_minmax_bw_kernel'),
                                note='bw_atomic_ms = tsamd_spmm_minmax_bw (bare spmm_min/max op): bounded by the device-scope '
                                     'atomic rate (20 G 64-byte segments/s, profiles/r02_ubench_atomics.csv), not by bytes'))
    par = dict()
    # backward: grad_mat against the fp64 formulas within the rounding bound of its arithmetic
    exact, bound = minmax_bw_bound(c, v, g, arg, n, dtype)
    err = (gmat.double() - exact).abs()
    par['grad_mat_max_err_over_bound'] = float((err / bound).max())
    par['grad_mat_autograd_equal_bound'] = float(((xr.grad.double() - exact).abs() / bound).max())
    par['grad_mat_atomic_route_max_err_over_bound'] = float(((gmat_a.double() - exact).abs() / bound).max())
    par['grad_mat_bare_op_autograd_max_err_over_bound'] = float(((xr2.grad.double() - exact).abs() / bound).max())
    par['grad_records_route_bit_identical_to_ids_route'] = records_same
    par['grad_mat_pull_deterministic'] = deterministic
    par['grad_mat_autograd_bit_identical_to_c_abi'] = same_as_op
    par['matmul_step_int32_ids_bit_identical_to_c_abi'] = train_same
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
        par['arg_out_mismatches'] = int((arg != ra.to(dev)).sum())  # compared on the device (ATen)
        par['out_bit_mismatches'] = int((out.view(torch.int16) != ro.to(dev).view(torch.int16)).sum())
        par['against'] = 'reference CPU kernel (forward, bit-exact); fp64 formulas of csrc/spmm.cpp:204-242 (backward)'
    par['ok'] = bool(par.get('arg_out_mismatches', 0) == 0 and par.get('out_bit_mismatches', 0) == 0 and
                     par['grad_mat_max_err_over_bound'] <= 1.0 and par['grad_mat_autograd_equal_bound'] <= 1.0 and
                     par['grad_mat_atomic_route_max_err_over_bound'] <= 1.0 and deterministic and same_as_op and
                     train_same and records_same and
                     par['grad_mat_bare_op_autograd_max_err_over_bound'] <= 1.0 and
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
    # row sums as segment sums over the CSR row pointers (fp64 ATen; index_add_ with hub rows serialises on its atomics)
    rpA, rpB = A.storage.rowptr(), B.storage.rowptr()

    def seg(v, rp):
        return torch.segment_reduce(v, 'sum', offsets=rp, initial=0.0)
    b_rowsum = seg(valB.double(), rpB)
    b_abs_rowsum = seg(valB.double().abs(), rpB)
    want_rows = seg(valA.double() * b_rowsum[colA], rpA)
    l1_rows = seg(valA.double().abs() * b_abs_rowsum[colA], rpA)
    # (segment sums over C's own rowptr: index_add_ on 1.3 G fp64 values with hub rows took 15 s per call on the device)
    rpC = C.storage.rowptr()
    got_rows = torch.segment_reduce(val.double(), 'sum', offsets=rpC, initial=0.0)
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
    # (two warm-up products: the second still allocates -- the result of the first is alive while it runs)
    ms = gpu_ms(lambda: A @ At, iters=iters, warm=2)
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
    if kind == 'c4':
        # what the reference itself does with GPU operands (torch_sparse/matmul.py:94-111): torch.sparse.mm on
        # device tensors = PyTorch-ROCm's hipSPARSE SpGEMM.  Context for a maintainer, not a target.
        try:
            Ad, Bd = A.to_torch_sparse_coo_tensor(), At.to_torch_sparse_coo_tensor()
            Cd = torch.sparse.mm(Ad, Bd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Cd = torch.sparse.mm(Ad, Bd)
            torch.cuda.synchronize()
            td = (time.perf_counter() - t0) * 1e3
            res['reference_gpu_route'] = dict(what='torch.sparse.mm on device COO tensors (PyTorch-ROCm hipSPARSE SpGEMM), '
                                                   'the call torch_sparse/matmul.py:104 makes for GPU inputs; 1 warm-up + 1 run',
                                              ms=round(td, 2), gproducts_per_s=round(P / td / 1e6, 3), nnz=int(Cd._nnz()),
                                              speedup_of_this_repo=round(td / ms, 2))
            del Ad, Bd, Cd
        except Exception as exc:  # noqa: BLE001
            res['reference_gpu_route'] = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        torch.cuda.empty_cache()
    if cpu:
        Ac, Bc = A.cpu().to_torch_sparse_coo_tensor(), At.cpu().to_torch_sparse_coo_tensor()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        Cc = torch.sparse.mm(Ac, Bc)  # what the reference calls (torch_sparse/matmul.py:104)
        t = time.perf_counter() - t0
        host_threads(32)
        res['cpu_baseline'] = dict(value=round(P / t / 1e9, 4), unit='GProducts/s', cores=cores, kind='reference',
                                   ms=round(t * 1e3, 1), sample='full workload, torch.sparse.mm on the host, 1 run')
        Cc = Cc.coalesce()
        row, col, val = C.coo()
        ci = Cc._indices().to(dev)  # whole-output comparison with ATen on the device
        idx_ok = bool(ci.size(1) == row.numel() and torch.equal(row, ci[0]) and torch.equal(col, ci[1]))
        ev = Cc._values().to(dev).double()
        # fp32 sums of ~1 product each, different order: 1e-5 relative to the L1 mass (>= |value|)
        l1 = (A.set_value(A.storage.value().abs(), 'coo') @ At.set_value(At.storage.value().abs(), 'coo')).storage.value()
        err = float(((val.double() - ev).abs() / l1.double().clamp(min=1e-30)).max()) if idx_ok else float('inf')
        res['parity'] = dict(against='torch.sparse.mm (CPU), whole output', nnz=int(nnzC), index_bit_exact=bool(idx_ok),
                             value_max_err_over_l1=err, tol_over_l1=1e-5, ok=bool(idx_ok and err <= 1e-5))
    return res


# ------------------------------------------------------------------------------------------------
# helpers of the rows below
# ------------------------------------------------------------------------------------------------
def wall_ms(fn, iters=5, warm=1):
    """Median host wall time of fn() with the device idle on both sides -- for calls that contain host
    syncs (data-dependent output sizes), where HIP events on the stream would miss the host part."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def _roof(nbytes, ms, scope):
    return dict(bound='hbm', algorithmic_bytes=int(nbytes), achieved=round(nbytes / ms / 1e6, 1), peak=HBM_PEAK_GBS,
                unit='GB/s', frac=round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), scope=scope)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# ------------------------------------------------------------------------------------------------
# C1: legacy torch_sparse.spmm(index, value, 1000, 1000, x) on 5 000 unsorted draws, F = 16 fp32
# ------------------------------------------------------------------------------------------------
def run_c1(dev, cpu=True, iters=50):
    """BASELINE.json configs[0] at its exact size.  Inputs and expected output come from the fixture
    tests/golden/py7_c1_spmm.npz, written by RUNNING the reference's torch_sparse/spmm.py:5-31
    (make_golden.py part7).  The CPU baseline is that pipeline (index_select * value, scatter_add)
    restated with ATen on the host, one thread (OpenMP fork/join dominates tiny inputs)."""
    import pytorch_sparse_amd as ts
    z = np.load(os.path.join(GOLDEN, 'py7_c1_spmm.npz'))
    m, n = int(z['m']), int(z['n'])
    index = torch.from_numpy(z['index']).to(dev)
    value = torch.from_numpy(z['value']).to(dev)
    x = torch.from_numpy(z['mat']).to(dev)
    E, K = index.size(1), x.size(1)
    fn = lambda: ts.spmm(index, value, m, n, x)  # noqa: E731
    ms_wall = wall_ms(fn, iters=iters, warm=3)
    ms_dev = gpu_ms(fn, iters=iters, warm=1)
    out = fn()
    ba = b_alg(E, m, K, 4, True, False) + E * 16  # + the (row, col) ids the sort reads once more
    res = dict(config='c1', workload='configs[0]: torch_sparse.spmm(index, value, 1000, 1000, x), %d unsorted COO draws '
                                     '(duplicates kept), F=%d fp32' % (E, K),
               dtype='f32', ms=round(ms_wall, 4), ms_device=round(ms_dev, 4), gedges_per_s=round(E / ms_wall / 1e6, 5),
               roofline=_roof(ba, ms_wall, 'whole call (order probe + host sync + radix sort + ind2ptr + SpMM), host wall '
                              'time: launch/sync latency bound at this size, the byte roofline does not apply'))
    ref, l1 = torch.from_numpy(z['out']).double(), torch.from_numpy(z['l1']).double()
    a = out.cpu().double()
    d = (a - ref).abs()
    bad = d > 1e-5 * ref.abs()
    par = dict(against='output of the reference Python torch_sparse.spmm on the same inputs (fixture py7_c1_spmm.npz); '
                       'criterion |gpu - ref| <= 1e-5 * sum|v x| everywhere and <= 1e-5 * |ref| where |ref| >= 0.1 sum|v x|',
               elements=int(a.numel()), max_err_over_l1=float((d / l1.clamp(min=1e-30)).max()), tol_over_l1=1e-5,
               max_rel_vs_ref=float((d / ref.abs().clamp(min=1e-30)).max()),
               n_rel_gt_1e_5_where_ref_ge_1e_1_l1=int((bad & (ref.abs() >= 1e-1 * l1)).sum()),
               n_rel_gt_1e_5_where_ref_ge_1e_2_l1=int((bad & (ref.abs() >= 1e-2 * l1)).sum()),
               vs_fp64_over_l1=float(((a - torch.from_numpy(z['out64'])).abs() / l1.clamp(min=1e-30)).max()))
    par['ok'] = bool(par['max_err_over_l1'] <= 1e-5 and par['vs_fp64_over_l1'] <= 1e-5 and
                     par['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0)
    res['parity'] = par
    if cpu:
        ic, vc, xc = index.cpu(), value.cpu(), x.cpu()
        torch.set_num_threads(1)

        def ref_pipeline():
            # torch_sparse/spmm.py:25-31, call for call: index_select, multiply, then torch_scatter.scatter_add
            # (third-party, not in the reference tree; scatter.py: broadcast the index to src's shape, zeros(...)
            # .scatter_add_(dim, index, src)).  No compiled reference code lies on this path: these ATen calls ARE it.
            o = xc.index_select(-2, ic[1])
            o = o * vc.unsqueeze(-1)
            idx = ic[0].unsqueeze(-1).expand_as(o)
            return torch.zeros(m, K).scatter_add_(-2, idx, o)
        t, ro, runs = cpu_time(ref_pipeline, budget_s=2.0, max_reps=200)
        res['cpu_baseline'] = dict(value=round(E / t / 1e9, 5), unit='GEdges/s', cores=1, kind='port', ms=round(t * 1e3, 4),
                                   sample='full workload; the ATen call sequence of torch_sparse/spmm.py:25-31 '
                                          '(index_select, mul, scatter_add_), one thread, best of %d' % runs)
        res['cpu_baseline']['port_matches_fixture'] = bool(torch.allclose(ro, torch.from_numpy(z['out']), rtol=1e-5, atol=1e-6))
        # BASELINE.json words configs[0] "via csrc/cpu reference": the same product through the COMPILED reference
        # kernels that travel to the GPU box (oracle/_ref: ind2ptr_cpu + spmm_cpu of csrc/cpu/convert_cpu.cpp:7-35,
        # spmm_cpu.cpp:8-101) behind a host sort by (row, col) -- the route SparseTensor.matmul takes in the reference.
        # When it is available THIS is the row's cpu_baseline (kind "reference"); the ATen port's time stays beside it.
        r = _ref_ops()
        if r is not None:
            def ref_route():
                perm = torch.argsort(ic[0] * n + ic[1], stable=True)
                rs, cs, vs = ic[0][perm], ic[1][perm], vc[perm]
                return r.spmm_sum(None, r.ind2ptr(rs, m), cs, vs, None, None, xc)
            t2, ro2, runs2 = cpu_time(ref_route, budget_s=2.0, max_reps=200)
            port = res['cpu_baseline']
            res['cpu_baseline'] = dict(
                value=round(E / t2 / 1e9, 5), unit='GEdges/s', cores=1, kind='reference', ms=round(t2 * 1e3, 4),
                sample='full workload; host argsort by (row, col) + the compiled reference ind2ptr_cpu + spmm_cpu '
                       '(oracle/_ref), one thread, best of %d' % runs2,
                reference_matches_fixture=bool(torch.allclose(ro2, torch.from_numpy(z['out']), rtol=1e-5, atol=1e-5)),
                port_ms=port['ms'], port_matches_fixture=port['port_matches_fixture'],
                port_note='the ATen call chain of torch_sparse/spmm.py:25-31 (index_select, mul, scatter_add_), same thread count')
    return res


# ------------------------------------------------------------------------------------------------
# construct / coalesce / transpose / t() on the 7.5 M-entry config-4 input
# ------------------------------------------------------------------------------------------------
def run_construct(dev, cpu=True, iters=5):
    """SURVEY 8a rows a9-a12 on BASELINE.json configs[3]'s input (500k x 500k, 7.5 M unsorted uniform draws):
    SparseTensor(row, col, value) (sort-on-construct + rowptr), functional coalesce, functional transpose and
    t() (csr2csc sort + permutation).  Index outputs are compared bit-for-bit with the numpy restatement of the
    reference Python (oracle/np_oracle.py, pinned to fixtures written by the reference itself); the CPU baseline
    is that pipeline (torch_sparse/storage.py:149-162, 431-466) restated with ATen on the host."""
    import pytorch_sparse_amd as ts
    from pytorch_sparse_amd import synth
    from oracle import np_oracle as npo
    m = n = 500000
    E = 7500000
    row, col = synth.uniform_edges(m, n, E, seed=0, device=dev)
    val = synth.values(E, device=dev)
    index = torch.stack([row, col])
    s = 4

    def ctor():
        A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n))
        A.storage.rowptr()
        return A
    A = ctor()

    def t_fresh():
        st = A.storage
        st._csr2csc = None
        st._csc2csr = None
        st._colptr = None
        st._colcount = None
        return A.t()
    ms = dict(construct=wall_ms(ctor, iters), coalesce=wall_ms(lambda: ts.coalesce(index, val, m, n), iters),
              transpose=wall_ms(lambda: ts.transpose(index, val, m, n), iters), t=wall_ms(t_fresh, iters))
    ci, cv = ts.coalesce(index, val, m, n)
    ti, tv = ts.transpose(index, val, m, n)
    At = t_fresh()
    E2 = ci.size(1)
    nbytes = dict(construct=E * (16 + s) + E * (16 + s) + (m + 1) * 8, coalesce=E * (16 + s) + E2 * (16 + s),
                  transpose=E * (16 + s) + E2 * (16 + s), t=E * (16 + s) + E * (16 + s) + E * 8)
    res = dict(config='construct', dtype='i64 ids + f32 values',
               workload='a9-a12 on the configs[3] input: %d unsorted COO draws over 500k x 500k (%d after coalescing)' % (E, E2),
               ms={k: round(v, 4) for k, v in ms.items()},
               mentries_per_s={k: round(E / v / 1e3, 1) for k, v in ms.items()},
               roofline={k: _roof(nbytes[k], ms[k], 'whole call incl. host syncs; bytes = E(16+s) in + E\'(16+s) out '
                                  '(SURVEY 8d; the radix passes are implementation cost)') for k in ms})
    res['ms_total'] = round(sum(ms.values()), 4)
    res['roofline']['construct'].update(pmc_traffic('construct_7m5', '*') or pmc_traffic('sort_coo_7m5', '*'))
    for k, w in (('coalesce', 'coalesce_7m5'), ('transpose', 'transpose_7m5')):
        res['roofline'][k].update(pmc_traffic(w, '*'))
    # ---- parity: every index output bit-exact against the numpy restatement, on the host ----
    rn, cn, vn = row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    rs, cs, perm = npo.sort_coo(rn, cn, m, n)
    ar, ac, av = A.coo()
    rowptr_ref = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(rs, minlength=m), out=rowptr_ref[1:])
    par = dict(against='oracle/np_oracle.py (reference Python restated, pinned to reference-written fixtures), whole outputs',
               entries=E,
               construct_row_col_bit_exact=bool(np.array_equal(ar.cpu().numpy(), rs) and np.array_equal(ac.cpu().numpy(), cs)),
               construct_value_bit_exact=bool(np.array_equal(av.cpu().numpy(), vn[perm])),
               construct_rowptr_bit_exact=bool(np.array_equal(A.storage.rowptr().cpu().numpy(), rowptr_ref)))
    orow, ocol, oval = npo.coalesce(rn, cn, vn, m, n)
    par['coalesce_index_bit_exact'] = bool(np.array_equal(ci.cpu().numpy(), np.stack([orow, ocol])))
    par['coalesce_value_max_abs_err'] = float(np.abs(cv.cpu().numpy().astype(np.float64) - oval.astype(np.float64)).max())
    par['coalesce_value_bit_exact'] = bool(np.array_equal(cv.cpu().numpy(), oval))
    trow, tcol, tval = npo.transpose(rn, cn, vn, m, n)
    par['transpose_index_bit_exact'] = bool(np.array_equal(ti.cpu().numpy(), np.stack([trow, tcol])))
    par['transpose_value_max_abs_err'] = float(np.abs(tv.cpu().numpy().astype(np.float64) - tval.astype(np.float64)).max())
    p2 = npo.csr2csc(rs, cs, m, n)
    tr, tc, tvv = At.coo()
    par['t_row_col_bit_exact'] = bool(np.array_equal(tr.cpu().numpy(), cs[p2]) and np.array_equal(tc.cpu().numpy(), rs[p2]))
    par['t_value_bit_exact'] = bool(np.array_equal(tvv.cpu().numpy(), vn[perm][p2]))
    par['csr2csc_bit_exact'] = bool(np.array_equal(A.storage.csr2csc().cpu().numpy(), p2))
    par['ok'] = bool(all(v for k, v in par.items() if k.endswith('bit_exact') and k != 'coalesce_value_bit_exact') and
                     par['coalesce_value_max_abs_err'] <= 1e-6 and par['transpose_value_max_abs_err'] <= 1e-6)
    res['parity'] = par
    if cpu:
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        rc, cc, vc = row.cpu(), col.cpu(), val.cpu()
        r = _ref_ops()

        def ref_construct():  # storage.py:149-162 + rowptr (storage.py:193)
            idx = rc * n + cc
            p = idx.argsort()
            r2, c2, v2 = rc[p], cc[p], vc[p]
            rp = r.ind2ptr(r2, m) if r is not None else torch._convert_indices_from_coo_to_csr(r2, m)
            return r2, c2, v2, rp

        def ref_coalesce():  # coalesce.py -> storage.py:149-162 + 431-466 (segment_csr = index_add_ over run ids)
            idx = rc * n + cc
            p = idx.argsort()
            ids = idx[p]
            mask = torch.ones_like(ids, dtype=torch.bool)
            mask[1:] = ids[1:] > ids[:-1]
            r2, c2 = rc[p][mask], cc[p][mask]
            seg = mask.cumsum(0) - 1
            v2 = torch.zeros(int(r2.numel()), dtype=vc.dtype).index_add_(0, seg, vc[p])
            return r2, c2, v2
        # (ATen's host sort / gathers do not scale to 256 threads: the better of a 32-thread and an all-core run)
        host_threads(32)
        t1, _, n1 = cpu_time(ref_construct, budget_s=6.0, max_reps=0)
        t2, _, n2 = cpu_time(ref_coalesce, budget_s=6.0, max_reps=0)
        host_threads(cores)
        t1b, _, _ = cpu_time(ref_construct, budget_s=6.0, max_reps=0)
        t2b, _, _ = cpu_time(ref_coalesce, budget_s=6.0, max_reps=0)
        if t1b < t1:
            t1, t2 = t1b, t2b
        else:
            cores = min(32, cores)
        res['cpu_baseline'] = dict(value=round(E / t1 / 1e6, 2), unit='Mentries/s (construct)', cores=cores, kind='port',
                                   ms=dict(construct=round(t1 * 1e3, 1), coalesce=round(t2 * 1e3, 1)),
                                   sample='full workload; torch_sparse/storage.py:149-162,431-466 restated with ATen on the '
                                          'host (argsort of row*N+col, gathers, ind2ptr / mask + index_add_), best of %d' % n1)
    return res


# ------------------------------------------------------------------------------------------------
# C2 backward: value gradient (SDDMM) and sum forward + both gradients through the autograd op
# ------------------------------------------------------------------------------------------------
def run_c2_backward(dev, cpu=True, iters=10):
    """SURVEY 8a rows a3 / a4 at config-2 size (2^20 R-MAT, F = 64 fp32): tsamd_spmm_value_bw alone, and
    adj.matmul(x) + backward(g) with both gradients through torch.ops.torch_sparse.spmm_sum.  CPU baseline and
    checker: the compiled reference's autograd op (csrc/spmm.cpp:55-113 over csrc/cpu/spmm_cpu.cpp:103-152 --
    its value gradient is single-threaded)."""
    import pytorch_sparse_amd as ts
    from pytorch_sparse_amd import synth
    from pytorch_sparse_amd import _native as nat
    rp, c, n = rmat_graph(20, 20, dev)
    E, K, s = c.numel(), 64, 4
    v = synth.values(E, device=dev)
    x = synth.features(n, K, device=dev)
    g = synth.features(n, K, seed=3, device=dev)
    A = ts.SparseTensor(rowptr=rp, col=c, value=v.clone().requires_grad_(), sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    A.storage.fill_cache_()
    row = A.storage.row()
    vb_ms = gpu_ms(lambda: nat.spmm_value_bw(row, rp, c, x, g, 'sum'), iters=iters)
    xr = x.clone().requires_grad_()

    def fwbw():
        xr.grad = None
        A.storage.value().grad = None
        out = A.matmul(xr, 'sum')
        out.backward(g)
        return out
    with operand_cache(False):
        fb_ms = gpu_ms(fwbw, iters=iters)  # steady state of a training loop: the pattern's CSC row ids are cached
        torch.ops.tsamd.pattern_cache(False)
        fb_nocache_ms = gpu_ms(fwbw, iters=iters)  # every backward reads (row, value) through csr2csc
        torch.ops.tsamd.pattern_cache(True)
    # the same step with FIXED edge weights (GCN's normalised adjacency: the values need no gradient): the
    # backward is A^T G alone and value[csr2csc] is cached next to the pattern's CSC row ids
    A_fixed = ts.SparseTensor(rowptr=rp, col=c, value=v, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    A_fixed.storage.fill_cache_()

    def fwbw_fixed():
        xr.grad = None
        A_fixed.matmul(xr, 'sum').backward(g)
    with operand_cache(False):
        fixed_ms = gpu_ms(fwbw_fixed, iters=iters)
    gmat_fixed = xr.grad.clone()
    out = fwbw()
    gval, gmat = A.storage.value().grad, xr.grad
    fixed_same = bool(torch.equal(gmat_fixed, gmat))
    b_vb = E * (16 + K * s + s) + n * K * s          # SURVEY 8d "value-grad SDDMM"
    b_fw = b_alg(E, n, K, s, True, False)
    b_gm = b_alg(E, n, K, s, True, False) + E * 8    # A^T G on the CSC view, entries read through csr2csc
    res = dict(config='c2_backward', dtype='f32',
               workload='configs[1] graph (2^20 R-MAT, E=%d), F=64 fp32: value gradient alone; sum forward + grad_value + '
                        'grad_mat through adj.matmul(x).backward(g)' % E,
               value_bw_ms=round(vb_ms, 4), fw_bw_ms=round(fb_ms, 4), fw_bw_without_pattern_cache_ms=round(fb_nocache_ms, 4),
               fw_bw_fixed_weights_ms=round(fixed_ms, 4), fixed_weights_grad_mat_bit_identical=fixed_same,
               gedges_per_s_value_bw=round(E / vb_ms / 1e6, 3),
               gedges_per_s_fw_bw=round(E / fb_ms / 1e6, 3),
               roofline=dict(_roof(b_vb, vb_ms, 'spmm_value_bw_kernel; bytes = E(16 + F s + s) + M F s (SURVEY 8d)'),
                             **pmc_traffic('c2_value_bw_f32_F64', 'spmm_value_bw_kernel')),
               roofline_fw_bw=_roof(b_fw + b_vb + b_gm, fb_ms, 'three kernels families of one training step: forward B_alg + '
                                    'value-grad bytes + B_alg of A^T G (+8 E for csr2csc)'))
    # independent fp64 yardsticks with ATen on the device (chunked gathers), none of the kernels under test
    ev = torch.empty(E, dtype=torch.float64, device=dev)
    l1v = torch.empty(E, dtype=torch.float64, device=dev)
    for a0 in range(0, E, 1 << 22):
        sl = slice(a0, min(E, a0 + (1 << 22)))
        xe, ge = x[c[sl]].double(), g[row[sl]].double()
        ev[sl] = (xe * ge).sum(1)
        l1v[sl] = (xe.abs() * ge.abs()).sum(1)
    l1v.clamp_(min=1e-30)
    par = dict(grad_value_vs_fp64_over_l1=float(((gval.double() - ev).abs() / l1v).max()), tol_over_l1=1e-5, elements_value=E)
    if cpu:
        r = _ref_ops()
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        st = A.storage
        rowc, rpc, cc, colptrc, permc = row.cpu(), rp.cpu(), c.cpu(), st.colptr().cpu(), st.csr2csc().cpu()
        vc, xc, gc = v.cpu(), x.cpu(), g.cpu()
        if r is not None:
            def ref_fwbw():
                vv, xx = vc.clone().requires_grad_(), xc.clone().requires_grad_()
                o = r.spmm_sum(rowc, rpc, cc, vv, colptrc, permc, xx)
                o.backward(gc)
                return o.detach(), vv.grad, xx.grad
            t, (ro, rgv, rgm), runs = cpu_time(ref_fwbw, budget_s=25.0, max_reps=1)
            res['cpu_baseline'] = dict(value=round(E / t / 1e9, 4), unit='GEdges/s (forward + both gradients)', cores=cores,
                                       kind='reference', ms=round(t * 1e3, 1),
                                       sample='full workload, compiled reference autograd op (value gradient serial, '
                                              'spmm_cpu.cpp:134-147), best of %d' % runs)
            par['against'] = ('compiled reference autograd op (csrc/spmm.cpp:55-113), whole grad_value [E] and grad_mat [N,F]; '
                              'criterion |gpu - ref| <= 1e-5 * L1 mass of the sum behind each element, ours and the reference\'s '
                              'also against fp64')
            par['grad_value_vs_ref_over_l1'] = float(((gval.cpu().double() - rgv.double()).abs() / l1v.cpu()).max())
            par['ref_grad_value_vs_fp64_over_l1'] = float(((rgv.double() - ev.cpu()).abs() / l1v.cpu()).max())
            row_t, v_t = rowc[permc], vc[permc]
            gm = sum_parity(gmat, colptrc, row_t, v_t, gc, rgm)
            par['grad_mat'] = {k: gm[k] for k in ('elements', 'max_err_over_l1', 'ours_vs_fp64_over_l1', 'ref_vs_fp64_over_l1',
                                                  'n_rel_gt_1e_5_where_ref_ge_1e_1_l1', 'n_rel_gt_1e_5_where_ref_ge_1e_2_l1',
                                                  'max_rel_vs_ref', 'ok')}
            fw = sum_parity(out.detach(), rpc, cc, vc, xc, ro)
            par['forward_max_err_over_l1'] = fw['max_err_over_l1']
            par['ok'] = bool(par['grad_value_vs_fp64_over_l1'] <= 1e-5 and par['grad_value_vs_ref_over_l1'] <= 1e-5 and
                             gm['ok'] and fw['ok'])
    if 'ok' not in par:
        par['against'] = 'fp64 ATen gathers on the device (grad_value only; no CPU leg in this run)'
        par['ok'] = bool(par['grad_value_vs_fp64_over_l1'] <= 1e-5)
    res['parity'] = par
    return res

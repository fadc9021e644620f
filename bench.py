"""bench.py -- headline benchmark of the sparse-matmul hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|c5] [--reduce sum]

One *step* = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: at N = 1 a CSR SpMM (tsamd_spmm: merge-path partition + merge + carry fix-up kernels); at
N > 1 each rank owns a row block of A and the matching row block of X, and a step is
"RCCL all-gather of X over xGMI, then SpMM on the local row block" (BASELINE.json north_star).
Scaling is WEAK: every rank owns 2**scale rows with ~edge_factor entries each, whatever N is, so
the global matrix is (N * 2**scale) square.

Workloads (default = the north-star shape the BASELINE.json target is
quoted on):
    ns   R-MAT scale 21, edge factor 20, F = 128 fp32     (default)
    c2   R-MAT scale 20, edge factor 20, F = 64  fp32     (BASELINE.json configs[1])
    c5   R-MAT scale 21, edge factor 32, F = 256 fp32     (per-GPU share of configs[4])

Rank 0 prints ONE JSON line (see README/DESIGN.md for the field meanings).  Timing: W warm-up
steps, then K steps between barrier + torch.cuda.synchronize() on both sides, max over ranks.
`roofline` is measured live on the SpMM merge kernel with HIP events on the launch stream
(tsamd_spmm_profiled); `cpu_baseline` times the reference's own CPU kernel (oracle/_ref, built
from /root/reference) on the host cores -- rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    'ns': dict(scale=21, edge_factor=20, F=128, desc='north-star: CSR SpMM 2M x 2M R-MAT ~20 nnz/row, F=128 fp32'),
    'c2': dict(scale=20, edge_factor=20, F=64, desc='configs[1]: CSR SpMM 1M x 1M R-MAT ~20 nnz/row, F=64 fp32'),
    'c5': dict(scale=21, edge_factor=32, F=256, desc='configs[4] per-GPU share: 2M rows x ~32 nnz/row, F=256 fp32'),
}


def b_alg(E, M, K, esize, has_value, minmax):
    """Algorithmic bytes of one SpMM (no-reuse gather model)."""
    return E * (8 + (esize if has_value else 0) + K * esize) + (M + 1) * 8 + M * K * esize + \
        (M * K * 8 if minmax else 0)


def local_block(scale, edge_factor, world, rank, device):
    """Rank-local row block: 2**scale rows, columns over all world * 2**scale vertices (R-MAT)."""
    from pytorch_sparse_amd import synth
    import math
    extra = int(math.log2(world)) if world > 1 else 0
    assert (1 << extra) == world, '--gpus must be a power of two'
    m = 1 << scale
    n = m * world
    row, col = synth.rmat_edges(scale, edge_factor, seed=1000 * rank, device=device)
    if extra:
        g = torch.Generator(device=device)
        g.manual_seed(77 + rank)
        # owner of each referenced column: uniform over the ranks (vertices are assumed to be
        # partitioned at random, the usual practice), the position inside the owner's block keeps
        # the R-MAT skew
        hi = torch.randint(0, world, (col.numel(), ), generator=g, device=device)
        col = hi * m + col
    rowptr, col = synth.to_csr(row, col, m, n)
    return rowptr, col, m, n


def cpu_baseline(rowptr, col, value, x, reduce, out=None):
    """The cpu_baseline leg -- the ONLY place bench.py touches oracle/: times the reference CPU kernel
    (oracle/_ref) on the host cores, bounded to <~ 30 s, and (given the GPU result `out`) spot-checks
    rows of the timed configuration against the C oracle."""
    res = _cpu_baseline_timing(rowptr, col, value, x, reduce)
    if out is not None and reduce in ('sum', 'mean'):
        worst = parity_sample(rowptr, col, value, x, out, reduce)
        res['parity'] = dict(rows_checked=50, max_err_over_l1=worst, tol=1e-5, ok=worst <= 1e-5)
    return res


def _cpu_baseline_timing(rowptr, col, value, x, reduce):
    cores = os.cpu_count() or 1
    rp, c, v, xx = rowptr.cpu(), col.cpu(), value.cpu(), x.cpu()
    E = c.numel()
    try:
        from oracle import ref
        have_ref = ref.available()
    except Exception:
        have_ref = False
    if have_ref:
        r = ref.ops()
        torch.set_num_threads(cores)
        fn = {'sum': lambda: r.spmm_sum(None, rp, c, v, None, None, xx),
              'mean': lambda: r.spmm_mean(None, rp, c, v, None, None, None, xx),
              'min': lambda: r.spmm_min(rp, c, v, xx), 'max': lambda: r.spmm_max(rp, c, v, xx)}[reduce]
        t0 = time.perf_counter()
        fn()
        first = time.perf_counter() - t0
        best = first
        reps = 0
        while reps < 3 and (reps + 2) * first < 25.0:
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
            reps += 1
        return dict(value=round(E / best / 1e9, 4), unit='GEdges/s', cores=cores, kind='reference',
                    sample='full workload (%d edges), compiled /root/reference csrc/cpu/spmm_cpu.cpp via '
                           'oracle/_ref, %d OpenMP threads, best of %d' % (E, cores, reps + 1),
                    ms=round(best * 1e3, 2))
    # scalar C port on a row sample
    from oracle import c_oracle as oc
    M = rp.numel() - 1
    ms = max(1, M // 16)
    e1 = int(rp[ms])
    t0 = time.perf_counter()
    oc.spmm(oc.F32, reduce, rp[:ms + 1].numpy(), c[:e1].numpy(), v[:e1].numpy(), xx.numpy())
    dt = time.perf_counter() - t0
    return dict(value=round(e1 / dt / 1e9, 4), unit='GEdges/s', cores=1, kind='port',
                sample='first 1/16 of the rows (%d edges), scalar C oracle' % e1, ms=round(dt * 1e3, 2))


def parity_sample(rowptr, col, value, x, out, reduce, nrows=48):
    """Spot-check rows of the timed configuration against the C oracle (fp32 rel 1e-5)."""
    import numpy as np
    from oracle import c_oracle as oc
    M = rowptr.numel() - 1
    deg = rowptr[1:] - rowptr[:-1]
    rows = torch.cat([torch.topk(deg, 2).indices.cpu(),
                      torch.randint(0, M, (nrows, ), generator=torch.Generator().manual_seed(0))]).unique()
    rp = rowptr[torch.stack([rows, rows + 1], 1).to(rowptr.device)].cpu()
    xc = x.cpu().numpy()
    worst = 0.0
    for i, r in enumerate(rows.tolist()):
        s, e = int(rp[i, 0]), int(rp[i, 1])
        c = col[s:e].cpu().numpy()
        v = value[s:e].cpu().numpy()
        ex, _ = oc.spmm(oc.F64, reduce, [0, e - s], c, v.astype(np.float64), xc.astype(np.float64))
        l1, _ = oc.spmm(oc.F64, 'sum', [0, e - s], c, np.abs(v).astype(np.float64), np.abs(xc).astype(np.float64))
        err = np.abs(out[r].cpu().double().numpy() - ex[0])
        worst = max(worst, float((err / (l1[0] + 1e-30)).max()))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='ns', choices=sorted(WORKLOADS))
    ap.add_argument('--reduce', default='sum', choices=['sum', 'mean', 'min', 'max'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--exchange', default='pipelined', choices=['pipelined', 'halo', 'allgather'],
                    help='N > 1: all_gather of X | halo = all_to_all of the referenced rows only | '
                         'pipelined = halo exchange in row pieces, overlapped with the SpMM of the previous piece')
    ap.add_argument('--chunks', type=int, default=8, help='row pieces of the pipelined exchange')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    from pytorch_sparse_amd import _native as nat
    from pytorch_sparse_amd import synth
    nat.lib()  # fail loudly if the HIP library is missing

    wl = WORKLOADS[args.workload]
    scale, ef, F = wl['scale'], wl['edge_factor'], wl['F']
    rowptr, col, m_local, n_global = local_block(scale, ef, world, rank, dev)
    E = col.numel()
    value = synth.values(E, seed=1 + rank, device=dev)
    x_local = synth.features(m_local, F, seed=2 + rank, device=dev)
    import pytorch_sparse_amd  # noqa: F401  (registers torch.ops.torch_sparse.*)
    from pytorch_sparse_amd.parallel import (HaloShardedSpMM, RowShardedSpMM, build_with_fallback,
                                             exchange_breakdown)

    def op_spmm(rp, c, v, x, reduce):
        # the drop-in path: the reference's own operator names, served by the HIP kernels
        if reduce == 'sum':
            return torch.ops.torch_sparse.spmm_sum(None, rp, c, v, None, None, x)
        if reduce == 'mean':
            return torch.ops.torch_sparse.spmm_mean(None, rp, c, v, None, None, None, x)
        if reduce == 'min':
            return torch.ops.torch_sparse.spmm_min(rp, c, v, x)[0]
        return torch.ops.torch_sparse.spmm_max(rp, c, v, x)[0]

    x_sizes = [m_local] * world

    # The requested exchange first; if its planning or a trial step raises on ANY rank, the ranks fall
    # back together (pipelined -> halo -> allgather) and the JSON line says so.
    requested = args.exchange
    sharded, args.exchange, fallback_reason = build_with_fallback(
        rowptr, col, value, x_sizes, x_local, args.reduce, op_spmm, requested, chunks=args.chunks,
        sync=torch.cuda.synchronize)
    comm_rows = getattr(sharded, 'n_needed', n_global) if world > 1 else 0

    def step():
        with torch.no_grad():
            return sharded(x_local, args.reduce)  # (RCCL exchange of X rows,) then the local SpMM

    # operands of ONE local SpMM launch, for the roofline / parity / cpu-baseline legs below
    if isinstance(sharded, RowShardedSpMM):
        x_full, col_k = sharded.gather(x_local), sharded.col
    elif isinstance(sharded, HaloShardedSpMM):
        x_full, col_k = sharded.exchange(x_local), sharded.col
    else:  # pipelined: reproduce the full block with a one-shot halo plan (setup only)
        ref_plan = HaloShardedSpMM(rowptr, col, value, x_sizes, None, op_spmm)
        x_full, col_k = ref_plan.exchange(x_local), ref_plan.col

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # N > 1: the exchange alone and the local SpMM alone, timed after the headline region, next to the
    # xGMI model of the exchange
    exchange_info = None
    if world > 1:
        exchange_info = exchange_breakdown(
            sharded, None if isinstance(sharded, (RowShardedSpMM, HaloShardedSpMM)) else ref_plan, x_local,
            lambda: op_spmm(rowptr, col_k, value, x_full, args.reduce), n_global, F * 4,
            reps=max(3, min(args.steps, 10)), sync=torch.cuda.synchronize)

    stats = torch.tensor([elapsed, float(E)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_edges = float(stats[1])
    else:
        total_edges = float(E)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gedges = total_edges * args.steps / elapsed / 1e9
        minmax = args.reduce in ('min', 'max')
        # ---- roofline of the dominant kernel, HIP events on the launch stream ----
        prof = []
        merge_ms = []
        for _ in range(10):
            nat.spmm(rowptr, col_k, value, x_full, args.reduce, profile=prof)
            merge_ms.append(prof[1])
        merge_ms.sort()
        k_ms = sum(merge_ms) / len(merge_ms)
        balg = b_alg(E, m_local, F, 4, True, minmax)
        achieved = balg / (k_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % args.workload)
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = dict(bound='hbm', kernel='tsamd::spmm_merge_kernel<float,4,ADD>', achieved=round(achieved, 1),
                        peak=HBM_PEAK_GBS, unit='GB/s', frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, algorithmic_bytes_per_launch=balg,
                        kernel_ms=round(k_ms, 4), partition_ms=round(prof[0], 4), fixup_ms=round(prof[2], 4),
                        whole_op_frac=round(balg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else None,
                        note='achieved = algorithmic (no-reuse) bytes / kernel time; it can exceed the HBM peak because '
                             'part of the gathered rows is served by L2 (compare traffic); whole_op_frac also counts '
                             'the relabel copy, partition and fix-up kernels of the same tsamd_spmm call')
        line = dict(metric='SpMM GEdges/s', value=round(gedges, 3), unit='GEdges/s', n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32',
                    data='synthetic',
                    config=dict(workload=wl['desc'], reduce=args.reduce, rows_per_gpu=m_local,
                                cols=n_global, edges_per_gpu=E, features=F,
                                graph='R-MAT(0.57,0.19,0.19,0.05) scale %d edge factor %d, coalesced' % (scale, ef),
                                parallelism='row-sharded x%d%s' % (world, (', RCCL %s of X rows (%d rows in per rank)' % ({'halo': 'all_to_all', 'pipelined': 'all_to_all in %d overlapped pieces' % args.chunks, 'allgather': 'all_gather'}[args.exchange], comm_rows)) if world > 1 else '')),
                    roofline=roofline)
        if exchange_info is not None:
            line['exchange'] = exchange_info
        if fallback_reason is not None:
            line['config']['exchange_fallback'] = 'requested %s; %s' % (requested, fallback_reason)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(rowptr, col_k, value, x_full, args.reduce, out)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

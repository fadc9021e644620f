"""bench.py -- headline benchmark of the sparse-matmul hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|c5] [--reduce sum] [--exchange allgather|halo|pipelined]

One *step* = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: at N = 1 a CSR SpMM (tsamd_spmm: merge-path partition + merge + carry fix-up kernels); at
N > 1 each rank owns a row block of A and the matching row block of X, and a step is
"RCCL all-gather of X over xGMI, then SpMM on the local row block" (BASELINE.json north_star).
Scaling is WEAK: every rank owns 2**scale rows with ~edge_factor entries each, whatever N is, so
the global matrix is (N * 2**scale) square.

Workloads:
    ns   R-MAT scale 21, edge factor 20, F = 128 fp32     (default at N = 1: the shape the BASELINE.json target is quoted on)
    c2   R-MAT scale 20, edge factor 20, F = 64  fp32     (BASELINE.json configs[1])
    c5   R-MAT scale 21, edge factor 32, F = 256 fp32     (default at N > 1: per-GPU share of configs[4]; the headline
                                                           step is the all-gather of X + the local SpMM, the halo and
                                                           pipelined exchanges are timed beside it)

Rank 0 prints ONE JSON line (see README/DESIGN.md for the field meanings).  Timing: W warm-up
steps, then K steps between barrier + torch.cuda.synchronize() on both sides, max over ranks.
`roofline` is measured live on the SpMM merge kernel with HIP events on the launch stream
(tsamd_spmm_profiled); `cpu_baseline` times the reference's own CPU kernel (oracle/_ref, built
from /root/reference) on the host cores -- rank 0, N = 1 only -- and its output is compared with
the WHOLE GPU output (`cpu_baseline.parity`).  At N = 1 the default run also carries
  `control`    the same kernel on a uniform-degree graph of the same size (load balance vs bandwidth)
  `secondary`  BASELINE.json configs C2 / C3 / C4 at their stated size, each with ms, roofline,
               cpu_baseline and whole-output parity (tests/baseline_configs.py; --no-secondary skips).
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

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    'ns': dict(scale=21, edge_factor=20, F=128, desc='north-star: CSR SpMM 2M x 2M R-MAT ~20 nnz/row, F=128 fp32'),
    'c2': dict(scale=20, edge_factor=20, F=64, desc='configs[1]: CSR SpMM 1M x 1M R-MAT ~20 nnz/row, F=64 fp32'),
    'c5': dict(scale=21, edge_factor=32, F=256, desc='configs[4] per-GPU share: 2M rows x ~32 nnz/row, F=256 fp32'),
}


def default_workload(world):
    """N = 1: the north-star shape; N > 1: BASELINE.json configs[4] (2^24 rows at 8 GPUs = 2^21 rows, ~32 nnz/row,
    F = 256 per rank), weak scaling, RCCL all-gather of X in front of the local SpMM."""
    return 'ns' if world == 1 else 'c5'


def b_alg(E, M, K, esize, has_value, minmax):
    """Algorithmic bytes of one SpMM (no-reuse gather model)."""
    return E * (8 + (esize if has_value else 0) + K * esize) + (M + 1) * 8 + M * K * esize + \
        (M * K * 8 if minmax else 0)


def b_min(E, M, N, K, esize, has_value, minmax):
    """Compulsory bytes: every operand read once, the result written once."""
    return E * (8 + (esize if has_value else 0)) + (M + 1) * 8 + N * K * esize + M * K * esize + \
        (M * K * 8 if minmax else 0)


def strong_block(scale, edge_factor, world, rank, device):
    """Strong scaling: ONE 2**scale square R-MAT matrix (the single-GPU workload, same seed on every
    rank), cut into `world` row blocks of equal nnz; X is sharded by the same row ranges.
    -> (rowptr, col, rows of this rank, columns, rows owned by every rank)."""
    from pytorch_sparse_amd import synth
    from pytorch_sparse_amd.parallel import narrow_rows, partition_rows
    n = 1 << scale
    rowptr, col = synth.rmat_csr(scale, edge_factor, seed=0, device=device)
    ranges = partition_rows(rowptr, world, 'nnz')
    s, e = ranges[rank]
    rp, c, _ = narrow_rows(rowptr, col, None, s, e)
    return rp.contiguous(), c.contiguous(), e - s, n, [b - a for a, b in ranges]


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


def cpu_baseline(rowptr, col, value, x, reduce, out):
    """The cpu_baseline leg -- the ONLY place bench.py touches oracle/ (through
    tests/baseline_configs.py): times the reference CPU kernel (oracle/_ref) on all host cores on the
    full workload and on ONE core on the first 1/32 of the rows, and compares its output with the
    whole GPU output."""
    from tests import baseline_configs as bc
    cores = os.cpu_count() or 1
    rp, c, v, xx = rowptr.cpu(), col.cpu(), value.cpu(), x.cpu()
    E, M = c.numel(), rp.numel() - 1
    torch.set_num_threads(cores)
    best, (ref_out, _, kind), runs = bc.cpu_time(lambda: bc.ref_spmm_cpu(rp, c, v, xx, reduce))
    res = dict(value=round(E / best / 1e9, 4), unit='GEdges/s', cores=cores, kind=kind, ms=round(best * 1e3, 2),
               sample='full workload (%d edges), %s, %d OpenMP threads, best of %d' % (
                   E, 'compiled /root/reference csrc/cpu/spmm_cpu.cpp via oracle/_ref' if kind == 'reference'
                   else 'C restatement oracle/ts_oracle.c', cores, runs))
    ms1 = M // 32
    e1 = int(rp[ms1])
    torch.set_num_threads(1)
    t1, _, _ = bc.cpu_time(lambda: bc.ref_spmm_cpu(rp[:ms1 + 1].contiguous(), c[:e1], v[:e1], xx, reduce),
                           budget_s=10.0, max_reps=1)
    torch.set_num_threads(cores)
    res['one_thread'] = dict(value=round(e1 / t1 / 1e9, 4), unit='GEdges/s', cores=1, ms=round(t1 * 1e3, 2),
                             sample='first 1/32 of the rows (%d edges)' % e1)
    if reduce == 'sum':
        res['parity'] = bc.sum_parity(out, rp, c, v, xx, ref_out)
        # the same product once more in the reference's ORDER OF OPERATIONS (tsamd_spmm_reference_order, a verification
        # mode of the library: csrc/spmm_ref_order.hip): every bit of the whole output against the compiled reference's
        if kind == 'reference':
            import pytorch_sparse_amd  # noqa: F401
            from pytorch_sparse_amd import _native as nat
            try:
                torch.ops.tsamd.reference_order(1)
                o2, _ = nat.spmm(rowptr, col, value, x, reduce)
                torch.cuda.synchronize()
            finally:
                torch.ops.tsamd.reference_order(0)
            ref_dev = ref_out.to(o2.device)
            res['parity']['reference_order_mode'] = dict(
                elements=int(o2.numel()),
                n_bits_differ=int((o2.view(torch.int32) != ref_dev.view(torch.int32)).sum()),
                note='tsamd_spmm_reference_order(1): one thread per output element, entries in CSR order, separately '
                     'rounded multiply and add -- bit-identical to csrc/cpu/spmm_cpu.cpp:61-87 by construction')
            del o2, ref_dev
    return res


def secondary(dev, cpu=True, stress=False):
    """BASELINE.json configs[0..3] at their stated size (tests/baseline_configs.py): C1 (legacy spmm), C2 forward, C2
    value-grad + sum forward/backward, C3 (value-less and with values, forward + backward), the configs[4] per-GPU share,
    construct / coalesce / transpose / t() on the C4 input, C4 -- each with ms, roofline, cpu_baseline and whole-output
    parity (statistics with ATen on the device); `--stress` adds the SpSpMM R-MAT stress row.  `wall_s` = what the row
    cost this run."""
    from tests import baseline_configs as bc
    legs = [lambda: bc.run_c1(dev, cpu=cpu), lambda: bc.run_c2(dev, cpu=cpu), lambda: bc.run_c2_backward(dev, cpu=cpu),
            lambda: bc.run_c3(dev, False, cpu=cpu), lambda: bc.run_c3(dev, True, cpu=cpu),
            lambda: bc.run_c5_share(dev, cpu=cpu, fp64_leg=False),
            lambda: bc.run_construct(dev, cpu=cpu), lambda: bc.run_spspmm(dev, 'c4', cpu=cpu)]
    if stress:  # SURVEY 8d stress row: property checks (the host SpGEMM would take minutes)
        legs.append(lambda: bc.run_spspmm(dev, 'stress', cpu=False, iters=3))
    res = []
    for fn in legs:
        t0 = time.perf_counter()
        try:
            row = fn()
        except Exception as exc:  # a failing secondary must not take the headline down
            row = dict(error='%s: %s' % (type(exc).__name__, exc))
        torch.cuda.synchronize()
        row['wall_s'] = round(time.perf_counter() - t0, 1)
        print('[bench] secondary %s: %.1f s' % (row.get('config', '?'), row['wall_s']), file=sys.stderr, flush=True)
        res.append(row)
        torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------------------
# the ONE stdout line: numbers only, a few KB (the driver parses the tail of stdout; round 4's 22 KB line did not
# fit its window).  Everything else -- notes, scopes, full parity statistics -- goes to stderr and to
# profiles/bench_last_full.json / gpurun_out/bench_last_full.json.
# ------------------------------------------------------------------------------------------------
def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + '~'


def compact_row(r):
    """One secondary row -> config, ms, frac, ok, cpu_ms (<= ~250 bytes)."""
    if 'error' in r:
        return dict(config=r.get('config', '?'), error=_short(r['error'], 80))
    o = dict(config=r['config'] + ('_val' if r.get('has_value') else ''))
    o.update(_pick(r, ('ms', 'fw_ms', 'bw_ms', 'bw_records_ms', 'bw_atomic_ms', 'matmul_fw_ms', 'matmul_fw_bw_ms', 'value_bw_ms', 'fw_bw_ms', 'fw_bw_fixed_weights_ms',
                       'gedges_per_s', 'gproducts_per_s')))
    roof = r.get('roofline') or {}
    if 'frac' in roof:
        o['frac'] = roof['frac']
    elif roof:  # construct: one roofline per call
        o['frac'] = {k: v.get('frac') for k, v in roof.items() if isinstance(v, dict)}
    for extra in ('roofline_bw', 'roofline_bw_records', 'roofline_fw_bw'):
        if isinstance(r.get(extra), dict) and 'frac' in r[extra]:
            o['frac_' + extra[9:]] = r[extra]['frac']
    cb = r.get('cpu_baseline')
    if cb:
        o['cpu_ms'] = cb.get('ms')
        o['cpu_kind'] = cb.get('kind')
    if isinstance(r.get('parity'), dict):
        o['ok'] = r['parity'].get('ok')
    if isinstance(r.get('reference_gpu_route'), dict) and 'ms' in r['reference_gpu_route']:
        o['hipsparse_ms'] = r['reference_gpu_route']['ms']
    return o


HEAD_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
             'vs_baseline', 'dtype', 'data')
ROOF_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'balg_over_peak',
             'algorithmic_bytes_per_launch', 'b_min', 'kernel_ms', 'pre_ms', 'fixup_ms', 'whole_op_balg_over_peak',
             'whole_op_frac')
MAX_LINE_BYTES = 6000


def compact_line(full):
    """The stdout line of a run from its full result: contract fields, `roofline`, `cpu_baseline`, `parity` of the
    headline, one small object per secondary row (and per exchange at N > 1).  No prose; < MAX_LINE_BYTES."""
    line = {k: (_short(full[k], 60) if isinstance(full[k], str) else full[k]) for k in HEAD_KEYS if k in full}
    cfg = full.get('config', {})
    line['config'] = {k: _short(v, 100) for k, v in _pick(cfg, ('workload', 'reduce', 'rows_per_gpu', 'cols', 'edges_per_gpu',
                                                                'features', 'parallelism', 'exchange_fallback')).items()}
    roof = _pick(full.get('roofline', {}), ROOF_KEYS)
    if 'traffic_source' in roof:
        src = roof['traffic_source']
        roof['traffic_source'] = 'this run: rocprofv3 --pmc, 2 passes' if src.startswith('this run') else _short(src, 40)
    line['roofline'] = roof
    cb = full.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind', 'ms'))
        line['cpu_baseline']['sample'] = _short(cb.get('sample', ''), 60)
        par = cb.get('parity')
        if par:
            p = _pick(par, ('ok', 'elements', 'max_err_over_l1', 'tol_over_l1', 'ours_vs_fp64_over_l1', 'ref_vs_fp64_over_l1',
                            'n_rel_gt_1e_5_where_ref_ge_1e_1_l1'))
            if 'reference_order_mode' in par:
                p['reference_order_n_bits_differ'] = par['reference_order_mode'].get('n_bits_differ')
            p.update(_pick(par.get('survey_8d_literal_bound', {}), ('n_viol_ours_vs_ref', 'n_viol_ours_vs_fp64',
                                                                    'n_viol_ref_vs_fp64')))
            line['parity'] = p
    for k in ('repeated_operand', 'relabelled_layout', 'control'):
        if isinstance(full.get(k), dict):
            line[k] = _pick(full[k], ('ms_per_step', 'ms', 'gedges_per_s'))
    ex = full.get('exchange')
    if isinstance(ex, dict):
        line['exchange'] = {k: v for k, v in ex.items() if isinstance(v, (int, float, bool))}
        if isinstance(ex.get('overlap'), dict):
            line['exchange']['overlap_parity_ok'] = (ex['overlap'].get('parity_vs_serial') or {}).get('ok')
    for k in ('exchange_variants_ms_per_step', 'extras'):
        if k in full:
            line[k] = _short(full[k], 160) if isinstance(full[k], str) else full[k]
    if isinstance(full.get('weak_scaling_reference'), dict):
        line['weak_scaling_reference'] = _pick(full['weak_scaling_reference'], ('gedges_per_s_per_gpu', ))
    if 'secondary' in full:
        line['secondary'] = [compact_row(r) for r in full['secondary']]
    if 'wall_s' in full:
        line['wall_s'] = full['wall_s']
    line['detail'] = 'profiles/bench_last_full.json'
    if len(json.dumps(line)) > MAX_LINE_BYTES:  # never let the explanatory part cost the headline
        for k in ('secondary', 'exchange', 'exchange_variants_ms_per_step', 'control', 'relabelled_layout', 'repeated_operand'):
            line.pop(k, None)
            if len(json.dumps(line)) <= MAX_LINE_BYTES:
                break
    return line


def emit(full):
    """Full result -> stderr + profiles/bench_last_full.json + gpurun_out/bench_last_full.json; compact line -> stdout,
    LAST (nothing may be printed to stdout after it)."""
    text = json.dumps(full)
    for d in ('profiles', 'gpurun_out'):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            with open(os.path.join(ROOT, d, 'bench_last_full.json'), 'w') as fh:
                fh.write(text + '\n')
        except OSError:
            pass
    print('[bench] full result: ' + text, file=sys.stderr, flush=True)
    print(json.dumps(compact_line(full)), flush=True)


def relabelled_leg(rowptr, col, value, x, out, reduce, steps, dev):
    """The same step with the dense operand KEPT in the relabelled layout (pytorch_sparse_amd/relabelled.py,
    tsamd_spmm_relabelled): what a multi-layer / multi-epoch caller pays per product.  Bit-identical values."""
    import pytorch_sparse_amd as ts
    m, n = rowptr.numel() - 1, x.size(0)
    A = ts.SparseTensor(rowptr=rowptr, col=col, value=value, sparse_sizes=(m, n), is_sorted=True, trust_data=True)
    t0 = time.perf_counter()
    x_h = ts.to_relabelled(x)
    ts.matmul_relabelled(A, x_h, reduce)  # fills the hashed-column cache
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3
    with torch.no_grad():
        for _ in range(3):
            ts.matmul_relabelled(A, x_h, reduce)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out_h = ts.matmul_relabelled(A, x_h, reduce)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    same = bool(torch.equal(ts.from_relabelled(out_h).view(torch.int32), out.view(torch.int32)))
    balg = b_alg(col.numel(), m, x.size(1), 4, True, reduce in ('min', 'max'))
    return dict(ms_per_step=round(ms, 4), gedges_per_s=round(col.numel() / ms / 1e6, 3),
                balg_over_peak=round(balg / ms / 1e6 / HBM_PEAK_GBS, 4), bit_identical_to_drop_in=same,
                one_time_setup_ms=round(setup_ms, 2),
                note='dense operand and result stay in the relabelled row order across products; the '
                     'drop-in op above re-copies X every call')


def control_graph(m, deg, F, dev, nat):
    """Uniform-degree control: exactly `deg` entries per row, uniform columns -- the same
    kernel without load imbalance, hub reuse or channel camping."""
    from pytorch_sparse_amd import synth
    from tests import baseline_configs as bc
    rp, c = synth.uniform_degree_csr(m, m, deg, seed=5, device=dev)
    v = synth.values(c.numel(), seed=6, device=dev)
    x = synth.features(m, F, seed=7, device=dev)
    ms = bc.gpu_ms(lambda: nat.spmm(rp, c, v, x, 'sum'), iters=10)
    ba = b_alg(c.numel(), m, F, 4, True, False)
    return dict(graph='uniform degree %d, uniform columns, %d rows' % (deg, m), edges=c.numel(), ms=round(ms, 4),
                gedges_per_s=round(c.numel() / ms / 1e6, 3), balg_over_peak=round(ba / ms / 1e6 / HBM_PEAK_GBS, 4))


def pmc_traffic_this_run(workload, reduce, kernel_substr='spmm_merge_kernel', timeout_s=150):
    """Fabric (L2 <-> Infinity Fabric) bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two short child
    runs of this script (`--pmc-child`: the headline steps only) under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` -- separate passes, counters only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (the two
    do not fit one pass; no trace domains next to --pmc).  bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in
    KiB): FETCH_SIZE counts the 128-byte fabric reads of 16 B/lane requests at 64 bytes on gfx950 (guide, HBM
    section); both factors are calibrated on this kernel family (profiles/traffic_ns.json: 2.000 / 1.000 on a
    streaming copy of known size, 1.009 x the algorithmic bytes on the no-reuse control graph).
    -> (bytes per launch, launches counted, description) or (None, 0, why not)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 0, 'rocprofv3 not found'
    vals = {}
    n_launch = 0
    tmp = tempfile.mkdtemp(prefix='tsamd_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out_dir = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--kernel-include-regex', kernel_substr, '--output-format', 'csv',
                   '-d', out_dir, '-o', 'pmc', '--', sys.executable, os.path.abspath(__file__), '--pmc-child',
                   '--workload', workload, '--reduce', reduce, '--steps', '3', '--warmup', '1']
            try:
                r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=timeout_s, text=True)
            except subprocess.TimeoutExpired:
                return None, 0, 'rocprofv3 --pmc %s timed out after %d s' % (counter, timeout_s)
            rows = []
            for root, _, files in os.walk(out_dir):
                for f in files:
                    if f.endswith('counter_collection.csv'):
                        with open(os.path.join(root, f)) as fh:
                            rows += [x for x in csv.DictReader(fh)
                                     if x.get('Counter_Name') == counter and kernel_substr in x.get('Kernel_Name', '')]
            if not rows:
                return None, 0, 'rocprofv3 --pmc %s produced no rows for %s (rc %d): %s' % (
                    counter, kernel_substr, r.returncode, (r.stdout or '')[-200:].replace('\n', ' '))
            # one row per dispatch (already summed over the XCDs / channels); a dispatch may appear once per dimension
            per = {}
            for x in rows:
                per[x['Dispatch_Id']] = per.get(x['Dispatch_Id'], 0.0) + float(x['Counter_Value'])
            vals[counter] = sum(per.values()) / len(per)
            n_launch = len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nbytes = (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
    return int(nbytes), n_launch, ('this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate counter-only passes over '
                                   '`bench.py --pmc-child`), mean of %d launches of %s; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB'
                                   % (n_launch, kernel_substr))


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment): re-run this command under
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 at a free port; rank 0's JSON line is the only
    thing on stdout.  On a box with fewer than N GPUs the ranks share devices and talk over gloo -- a REHEARSAL of
    the control flow that the line's `data` field flags as not a measurement (TSAMD_BENCH_BACKEND=gloo, which can
    also be set by hand)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and 'TSAMD_BENCH_BACKEND' not in env:
        print('[bench] %d GPU(s) visible, %d ranks asked for: rehearsal over gloo, ranks share devices' % (have, n),
              file=sys.stderr)
        env['TSAMD_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# what main() leaves for its wrapper once the headline of an N > 1 run is settled: the bare line, the rank, the "printed" flag
_SETTLED = {}


def main():
    """`_main()`; N > 1 only: if an explanatory leg raises after the headline was settled (they have never run on real
    xGMI hardware), rank 0 still prints the headline line -- with the exception in `extras` -- instead of dying silently,
    and the process leaves with status 0 (its peers are ended by their watchdogs)."""
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001
        if not _SETTLED:
            raise
        import traceback
        traceback.print_exc()
        if _SETTLED['rank'] == 0 and not _SETTLED['printed'].is_set():
            line = dict(_SETTLED['bare'])
            line['extras'] = 'withheld: an explanatory leg after the timed region raised %s: %s' % (
                type(exc).__name__, str(exc)[:300])
            print(json.dumps(compact_line(line)), flush=True)
        os._exit(0)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help='default: ns at N = 1 (the north-star shape the BASELINE target is quoted on), c5 at N > 1 '
                         '(the per-GPU share of BASELINE.json configs[4])')
    ap.add_argument('--reduce', default='sum', choices=['sum', 'mean', 'min', 'max'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the C2 / C3 / C4 entries (N = 1 only)')
    ap.add_argument('--stress', action='store_true', help='add the SpSpMM R-MAT stress row to the secondary rows (~40 s)')
    ap.add_argument('--headline-only', action='store_true',
                    help='only the timed north-star steps + the roofline launches (for rocprofv3: every launch of the '
                         'dominant kernel in the trace is then the headline workload)')
    ap.add_argument('--no-pmc', action='store_true',
                    help='do not spawn the two rocprofv3 --pmc child runs that measure the fabric traffic of the dominant '
                         'kernel (N = 1); roofline.traffic then comes from profiles/traffic_<workload>.json and says so')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)  # the child run under rocprofv3 --pmc
    ap.add_argument('--exchange', default='allgather', choices=['pipelined', 'halo', 'allgather', 'allgather_serial'],
                    help='N > 1, the exchange of the HEADLINE step: allgather (default: the north star names the RCCL '
                         'all-gather of X; sent in the camping-free row order in --ag-chunks collectives, each overlapped '
                         'with the partial product of the column block that landed before, SURVEY 8e) | allgather_serial = '
                         'one all-gather, then the drop-in op | halo = all_to_all of the referenced rows only | pipelined = '
                         'halo exchange in row pieces, overlapped with the SpMM of the previous piece.  The others are '
                         'timed beside it (exchange_variants_ms_per_step)')
    ap.add_argument('--chunks', type=int, default=8, help='row pieces of the pipelined exchange')
    ap.add_argument('--ag-chunks', type=int, default=4, help='collectives (row chunks of every shard) of the overlapped all-gather')
    ap.add_argument('--extras-timeout', type=int, default=420,
                    help='N > 1: seconds the legs after the timed region may take before the headline line is printed '
                         'without them and the ranks are ended')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak (default): every rank owns 2**scale rows; strong: the single-GPU matrix is cut '
                         'into N row blocks of equal nnz')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.workload = args.workload or default_workload(world)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        sys.exit(self_launch(args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback exists)'
    # TSAMD_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices, collectives go through gloo) -- a functional check, never a measurement
    backend = os.environ.get('TSAMD_BENCH_BACKEND', 'nccl')
    local_rank = local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from pytorch_sparse_amd import _native as nat
    from pytorch_sparse_amd import synth
    nat.lib()  # fail loudly if the HIP library is missing

    wl = WORKLOADS[args.workload]
    scale, ef, F = wl['scale'], wl['edge_factor'], wl['F']
    if args.scaling == 'strong' and world > 1:
        rowptr, col, m_local, n_global, x_sizes = strong_block(scale, ef, world, rank, dev)
    else:
        rowptr, col, m_local, n_global = local_block(scale, ef, world, rank, dev)
        x_sizes = [m_local] * world
    E = col.numel()
    value = synth.values(E, seed=1 + rank, device=dev)
    x_local = synth.features(m_local, F, seed=2 + rank, device=dev)
    import pytorch_sparse_amd  # noqa: F401  (registers torch.ops.torch_sparse.*)
    from pytorch_sparse_amd.parallel import (HaloShardedSpMM, OverlappedAllGatherSpMM, RowShardedSpMM,
                                             build_with_fallback, exchange_breakdown)

    def op_spmm(rp, c, v, x, reduce):
        # the drop-in path: the reference's own operator names, served by the HIP kernels
        if reduce == 'sum':
            return torch.ops.torch_sparse.spmm_sum(None, rp, c, v, None, None, x)
        if reduce == 'mean':
            return torch.ops.torch_sparse.spmm_mean(None, rp, c, v, None, None, None, x)
        if reduce == 'min':
            return torch.ops.torch_sparse.spmm_min(rp, c, v, x)[0]
        return torch.ops.torch_sparse.spmm_max(rp, c, v, x)[0]

    # The requested exchange first; if its planning or a trial step raises on ANY rank, the ranks fall
    # back together (pipelined -> halo -> allgather) and the JSON line says so.
    requested = args.exchange
    sharded, args.exchange, fallback_reason = build_with_fallback(
        rowptr, col, value, x_sizes, x_local, args.reduce, op_spmm, requested, chunks=args.chunks,
        sync=torch.cuda.synchronize, ag_chunks=args.ag_chunks)
    comm_rows = getattr(sharded, 'n_needed', n_global) if world > 1 else 0

    from pytorch_sparse_amd.parallel import PipelinedHaloSpMM as _Pipelined

    def run_op(op):
        # inference steps: the pipelined plan is told so (otherwise it agrees on the path with one all_reduce per call)
        if isinstance(op, (_Pipelined, OverlappedAllGatherSpMM)):
            return op(x_local, args.reduce, differentiable=False)
        return op(x_local, args.reduce)

    def step():
        with torch.no_grad():
            return run_op(sharded)  # (RCCL exchange of X rows,) then the local SpMM

    # operands of ONE local SpMM launch, for the roofline / parity / cpu-baseline legs below
    if isinstance(sharded, (RowShardedSpMM, OverlappedAllGatherSpMM)):
        x_full, col_k = RowShardedSpMM(rowptr, col, value, x_sizes, None, op_spmm).gather(x_local), col
    elif isinstance(sharded, HaloShardedSpMM):
        x_full, col_k = sharded.exchange(x_local), sharded.col
    else:  # pipelined: reproduce the full block with a one-shot halo plan (setup only)
        ref_plan = HaloShardedSpMM(rowptr, col, value, x_sizes, None, op_spmm)
        x_full, col_k = ref_plan.exchange(x_local), ref_plan.col

    # The headline steps run with the operand cache OFF (the ops' default: they keep no state between calls): every
    # step does all of its work (probe, relabelled copy of X, partition, merge, fix-up).  What a caller sees who
    # OPTS IN to the cache and multiplies by the same X again is measured separately below (`repeated_operand`).
    torch.ops.tsamd.operand_cache(False)
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
    if args.pmc_child:  # under rocprofv3 --pmc: the headline launches are all that is wanted
        return

    # the headline number is settled HERE (max over ranks of the timed region, edges summed over ranks), before any of
    # the explanatory legs below issues another collective
    stats = torch.tensor([elapsed, float(E)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_edges = float(stats[1])
    else:
        total_edges = float(E)
    # N > 1: everything after this point (exchange breakdown, the other exchanges, roofline launches) only EXPLAINS the
    # number.  If one of those legs stalls on a collective -- they have never run on real xGMI hardware -- a watchdog
    # prints the headline line without them after --extras-timeout seconds and ends every rank, so that an unattended
    # `bench.py --gpus 8` always yields its one JSON line.
    import threading
    watchdog_done, line_printed = None, threading.Event()
    if world > 1:
        watchdog_done = threading.Event()
        ms0 = elapsed / args.steps * 1e3
        bare = dict(metric='SpMM GEdges/s', value=round(total_edges * args.steps / elapsed / 1e9, 3), unit='GEdges/s',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms0, 4), higher_is_better=True,
                    scaling=args.scaling, vs_baseline=None, dtype='f32',
                    data='synthetic' if backend == 'nccl' else 'synthetic (REHEARSAL over %s, ranks share GPUs: not a measurement)' % backend,
                    config=dict(workload=wl['desc'], reduce=args.reduce, rows_per_gpu=m_local, cols=n_global,
                                edges_per_gpu=E, features=F, parallelism='row-sharded x%d, exchange %s' % (world, args.exchange)),
                    extras='withheld: the explanatory legs after the timed region did not finish within %d s'
                           % args.extras_timeout)

        def watchdog():
            if watchdog_done.wait(args.extras_timeout):
                return
            if rank == 0 and not line_printed.is_set():
                print(json.dumps(compact_line(bare)), flush=True)
            os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        _SETTLED.update(bare=bare, rank=rank, printed=line_printed)

    # N = 1: the same steps with the operand cache opted IN (torch.ops.tsamd.operand_cache(True)): the second and
    # later calls with an unchanged X find its relabelled copy (tsamd_spmm_cached) -- bit-identical output
    repeated = None
    if world == 1:
        torch.ops.tsamd.operand_cache(True)
        with torch.no_grad():
            first = step()
            same = bool(torch.equal(first.view(torch.int32), out.view(torch.int32)))
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ms_rep = (time.perf_counter() - t1) / args.steps * 1e3
        _, hits, fills = torch.ops.tsamd.operand_cache(True)
        repeated = dict(ms_per_step=round(ms_rep, 4), gedges_per_s=round(E / ms_rep / 1e6, 3), hits=int(hits), fills=int(fills),
                        bit_identical_to_headline=same,
                        note='OPT-IN operand cache (torch.ops.tsamd.operand_cache(True); off by default since a sparse write '
                             'through x.data is invisible to it): calls after the first with the same X (same storage, version '
                             'counter, shape, stream, pattern; sampled fingerprint re-checked on the device) skip '
                             'spmm_probe_kernel and spmm_permute_rows_kernel.  NOT the headline: the headline steps above run '
                             'with the cache off')
    torch.ops.tsamd.operand_cache(False)

    # N > 1: the exchange alone and the local SpMM alone, timed after the headline region, next to the
    # xGMI model of the exchange
    exchange_info = None
    if world > 1:
        staged = isinstance(sharded, OverlappedAllGatherSpMM)
        if staged:  # the compute of the overlapped step alone: the column-block stages on buffers that have landed
            x_pad = sharded.wire_order(x_local)
            landed = sharded.gather_all(x_local)
            spmm_alone = lambda: sharded.multiply_landed(x_pad, landed, args.reduce)  # noqa: E731
        else:
            spmm_alone = lambda: op_spmm(rowptr, col_k, value, x_full, args.reduce)  # noqa: E731
        exchange_info = exchange_breakdown(
            sharded, None if isinstance(sharded, (RowShardedSpMM, HaloShardedSpMM, OverlappedAllGatherSpMM)) else ref_plan,
            x_local, spmm_alone, n_global, F * 4, reps=max(3, min(args.steps, 10)), sync=torch.cuda.synchronize)
        if staged:
            # parity of the overlapped step against the serial one (one all-gather, then the drop-in op) on this rank
            serial = op_spmm(rowptr, col, value, x_full, args.reduce)
            l1 = op_spmm(rowptr, col, value.abs(), x_full.abs(), 'sum') if args.reduce in ('sum', 'mean') else None
            if l1 is not None:
                worst = float(((out.double() - serial.double()).abs() / l1.double().clamp(min=1e-30)).max()) / \
                    (1.0 if args.reduce == 'sum' else 1.0)
                okp = worst <= 1e-5 if args.reduce == 'sum' else bool(torch.allclose(out, serial, rtol=1e-5, atol=1e-5))
            else:
                worst = float((out != serial).sum())
                okp = worst == 0
            flag = torch.tensor([0.0 if okp else 1.0, worst], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            exchange_info['overlap'] = dict(
                chunks=args.ag_chunks, stages=len(sharded.stages), rows_per_chunk=sharded.cs,
                staged_spmm_only_ms=exchange_info['spmm_only_ms'],
                parity_vs_serial=dict(ok=bool(float(flag[0]) == 0.0), worst=float(flag[1]),
                                      criterion='sum: |staged - serial| <= 1e-5 * sum_e|v_e x_e| for every element of every '
                                                'rank; min / max: bit-identical'),
                note='all-gather of X in %d collectives of rows in the camping-free wire order; the partial product of '
                     'the columns a collective delivered runs while the next one is in flight (tsamd_spmm_partial), the '
                     'rank\'s own column block first; exposed_ms = step - staged_spmm_only_ms' % args.ag_chunks)
            del serial, l1, landed, x_pad

    # N > 1: the same step with each of the three exchanges (the north star names the all-gather; the
    # halo variants move only the referenced rows), a few steps each, max over ranks
    variants = None
    if world > 1:
        variants = {}
        from pytorch_sparse_amd.parallel import EXCHANGES, PipelinedHaloSpMM
        for mode in ('allgather', 'allgather_serial', 'halo', 'pipelined'):
            try:
                if mode == args.exchange:
                    op_v = sharded
                else:
                    kw = dict(chunks=args.chunks) if EXCHANGES[mode] is PipelinedHaloSpMM else {}
                    if EXCHANGES[mode] is OverlappedAllGatherSpMM:
                        kw = dict(chunks=args.ag_chunks)
                    op_v = EXCHANGES[mode](rowptr, col, value, x_sizes, None, op_spmm, **kw)
                reps = max(3, min(args.steps, 10))
                with torch.no_grad():
                    run_op(op_v)
                    torch.cuda.synchronize()
                    dist.barrier()
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        run_op(op_v)
                    torch.cuda.synchronize()
                    dist.barrier()
                    ms_v = (time.perf_counter() - t1) / reps * 1e3
                failed = 0
            except Exception:  # noqa: BLE001
                ms_v, failed = 0.0, 1
            tv = torch.tensor([ms_v, float(failed)], dtype=torch.float64, device=dev)
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
            variants[mode] = None if float(tv[1]) > 0 else round(float(tv[0]), 4)
            if mode != args.exchange:
                del op_v
                torch.cuda.empty_cache()

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
        # fabric traffic of the dominant kernel: measured now (two rocprofv3 --pmc child runs of the headline steps);
        # only when that is impossible, the committed summary of an earlier builder run -- and the line says which
        traffic, traffic_source, pmc_why = None, None, None
        if world == 1 and not args.no_pmc:
            traffic, _, traffic_source = pmc_traffic_this_run(args.workload, args.reduce)
            if traffic is None:
                pmc_why, traffic_source = traffic_source, None
        tfile = os.path.join(ROOT, 'profiles', 'traffic_%s.json' % args.workload)
        # (the committed figure is the WHOLE single-GPU workload's: a rank's shard of it at N > 1 moves other bytes)
        if traffic is None and world == 1 and os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_source = 'profiles/traffic_%s.json (%s), NOT measured in this run%s' % (
                    args.workload, tj.get('source', 'rocprofv3 --pmc, builder run'),
                    (' (in-run measurement failed: %s)' % pmc_why) if pmc_why else '')
            except Exception:
                traffic = None
        bmin = b_min(E, m_local, n_global, F, 4, True, minmax)
        # `achieved` / `frac` are PHYSICAL: bytes that crossed the L2 <-> fabric boundary (counters) / kernel time.  The
        # algorithmic (no-reuse) byte model of SURVEY 8d over-counts the hub rows that hit in L2 and exceeds the
        # peak on this graph: it is reported as balg_over_peak, a throughput index, never as the bandwidth share.
        phys = traffic if traffic else None
        achieved_phys = (phys / (k_ms * 1e-3) / 1e9) if phys else None
        roofline = dict(bound='hbm', kernel='tsamd::spmm_merge_kernel<float,4,ADD>',
                        achieved=round(achieved_phys, 1) if phys else round(min(achieved, HBM_PEAK_GBS), 1),
                        peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved_phys / HBM_PEAK_GBS, 4) if phys else None,
                        frac_basis=('fabric traffic (PMC counters) / kernel time / peak' if phys else
                                    'no counter traffic available: frac withheld (balg_over_peak is not a bandwidth share)'),
                        traffic=traffic, traffic_source=traffic_source,
                        balg_over_peak=round(achieved / HBM_PEAK_GBS, 4), algorithmic_gbs=round(achieved, 1),
                        algorithmic_bytes_per_launch=balg, b_min=bmin,
                        traffic_over_b_min=(round(traffic / bmin, 2) if traffic else None),
                        hbm_bytes_bounds=[bmin, traffic] if traffic else None,
                        streaming_ceiling_gbs=6300.0,
                        frac_of_streaming_ceiling=round(achieved_phys / 6300.0, 4) if phys else None,
                        kernel_ms=round(k_ms, 4), pre_ms=round(prof[0], 4), fixup_ms=round(prof[2], 4),
                        whole_op_balg_over_peak=round(balg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else None,
                        whole_op_frac=(round((traffic + 2 * (n_global * F * 4)) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                       if (traffic and world == 1) else None),
                        status='saturated: this kernel runs at the streaming ceiling of the machine on the no-reuse control '
                               'graph (`control`) and at >= 0.9 of the 8 TB/s peak in fabric bytes here; what is left at the '
                               'north star is traffic_over_b_min (re-fetches of gathered rows), not kernel speed '
                               '(2-D blocking measured negative, profiles/r02_exp_colblock.jsonl)',
                        note='frac = measured fabric bytes per launch / HIP-event kernel time / 8 TB/s.  Fabric bytes include '
                             'Infinity-Cache hits (the TCC_EA0 counters sit in front of the MALL; rocprofv3 -L on this stack '
                             'lists no MALL / HBM-side counter, TCC_EA0_*_DRAM only tells DRAM from GMI / IO targets), so the '
                             'HBM bytes proper lie in hbm_bytes_bounds = [compulsory b_min, fabric traffic]; that is how frac '
                             'can exceed the 6.3 TB/s streaming ceiling.  balg_over_peak = ALGORITHMIC (no-reuse gather '
                             'model, SURVEY 8d) bytes / kernel time / peak: a throughput index that exceeds 1 when gathered '
                             'rows hit in L2.  pre_ms = probe + relabelled copy of X + merge-path partition, fixup_ms = carry '
                             'fix-up; whole_op_frac adds the copy\'s 2 N F s bytes and divides by the whole step')
        line = dict(metric='SpMM GEdges/s', value=round(gedges, 3), unit='GEdges/s', n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                    higher_is_better=True, scaling=(args.scaling if world > 1 else 'weak'), vs_baseline=None, dtype='f32',
                    data='synthetic' if backend == 'nccl' else 'synthetic (REHEARSAL over %s, ranks share GPUs: not a measurement)' % backend,
                    config=dict(workload=wl['desc'], reduce=args.reduce, rows_per_gpu=m_local,
                                cols=n_global, edges_per_gpu=E, features=F,
                                graph='R-MAT(0.57,0.19,0.19,0.05) scale %d edge factor %d, coalesced' % (scale, ef),
                                parallelism='row-sharded x%d%s' % (world, (', RCCL %s of X rows (%d rows in per rank)' % ({'halo': 'all_to_all', 'pipelined': 'all_to_all in %d overlapped pieces' % args.chunks, 'allgather': 'all_gather in %d chunks overlapped with column-block partial products' % args.ag_chunks, 'allgather_serial': 'all_gather, then SpMM'}[args.exchange], comm_rows)) if world > 1 else '')),
                    roofline=roofline)
        if exchange_info is not None:
            exchange_info['exposed_ms'] = round(max(0.0, ms_per_step - exchange_info['spmm_only_ms']), 3)
            # what the mandated design can reach at best with PERFECT overlap: the step cannot be shorter than the longer
            # of the local product and the modelled exchange (print the expectation beside the measurement)
            mod = exchange_info.get('modelled_exchange_ms')
            if mod:
                exchange_info['modelled_efficiency'] = round(
                    exchange_info['spmm_only_ms'] / max(exchange_info['spmm_only_ms'], mod), 4)
                exchange_info['measured_efficiency'] = round(exchange_info['spmm_only_ms'] / ms_per_step, 4)
            line['exchange'] = exchange_info
            # the N = 1 default of this script is ANOTHER workload (ns); the single-GPU rate of THIS workload -- the
            # reference point of a weak-scaling efficiency -- is the local SpMM without the exchange:
            line['weak_scaling_reference'] = dict(
                gedges_per_s_per_gpu=round(E / exchange_info['spmm_only_ms'] / 1e6, 3),
                note='one rank\'s block of this workload multiplied without any exchange (spmm_only_ms); compare '
                     'value / n_gpus with this, not with the N = 1 default run (workload ns); `python bench.py --gpus 1 '
                     '--workload c5` measures the same thing stand-alone')
        if variants is not None:
            line['exchange_variants_ms_per_step'] = variants
            line['exchange_variants_gedges_per_s'] = {k: (None if v is None else round(total_edges / v / 1e6, 3))
                                                      for k, v in variants.items()}
        if fallback_reason is not None:
            line['config']['exchange_fallback'] = 'requested %s; %s' % (requested, fallback_reason)
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            line['cpu_baseline'] = cpu_baseline(rowptr, col_k, value, x_full, args.reduce, out)
        if repeated is not None:
            line['repeated_operand'] = repeated
        if world == 1 and not args.headline_only:
            line['relabelled_layout'] = relabelled_leg(rowptr, col_k, value, x_full, out, args.reduce,
                                                       max(10, min(args.steps, 50)), dev)
        if world == 1 and not args.no_secondary and not args.headline_only:
            del x_full, out, sharded
            torch.cuda.empty_cache()
            line['control'] = control_graph(m_local, ef, F, dev, nat)
            line['secondary'] = secondary(dev, cpu=not args.no_cpu_baseline, stress=args.stress)
        line['wall_s'] = round(time.perf_counter() - T_START, 1)
        line_printed.set()
        emit(line)
    if world > 1:
        dist.barrier()
    if watchdog_done is not None:  # (armed through the last barrier: a peer that stalled must not keep the others)
        watchdog_done.set()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

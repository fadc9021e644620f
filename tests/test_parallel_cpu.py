"""world_size-2 gloo tests of the row-sharded SpMM path (sharding, all-gather, reduce-scatter
logic).  The local multiply is injected: the C oracle for the forward parity check, a
differentiable torch formulation for the backward check.  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle as oc
from pytorch_sparse_amd import synth
from pytorch_sparse_amd.parallel import narrow_rows, partition_rows


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_spmm(rowptr, col, value, x, reduce):
    o, _ = oc.spmm(oc.F32, reduce, rowptr.numpy(), col.numpy(), None if value is None else value.numpy(),
                   x.detach().numpy())
    return torch.from_numpy(o)


def torch_spmm_sum(rowptr, col, value, x, reduce):
    """Differentiable torch formulation of the local product (sum, and min / max through
    scatter_reduce, whose backward routes the gradient to the winning entry like csrc/spmm.cpp:204-242)."""
    M = rowptr.numel() - 1
    row = torch.repeat_interleave(torch.arange(M), rowptr[1:] - rowptr[:-1])
    prod = value[:, None] * x[col]
    if reduce in ('sum', 'mean'):
        tot = torch.zeros(M, x.size(1), dtype=x.dtype).index_add_(0, row, prod)
        if reduce == 'mean':
            tot = tot / (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(x.dtype)[:, None]
        return tot
    assert reduce in ('min', 'max')
    return torch.zeros(M, x.size(1), dtype=x.dtype).scatter_reduce(
        0, row[:, None].expand_as(prod), prod, reduce='a' + reduce, include_self=False)


def _worker(rank, world, port, balance, q, exchange='allgather'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pytorch_sparse_amd.parallel import shard_matrix
        rp, c = synth.rmat_csr(9, 8, seed=1)
        n = 1 << 9
        v = synth.values(c.numel())
        x = synth.features(n, 12)
        # forward parity for every reduction, local multiply = C oracle
        kw = {}
        if exchange == 'allgather':  # the overlapped all-gather: column-block partial products, combined by the
            from tests.util import ref_minmax_bw, ref_partial, ref_value_bw  # restated contracts of tsamd_spmm_partial /
            kw = dict(chunks=3, partial_fn=ref_partial, value_bw_fn=ref_value_bw, minmax_bw_fn=ref_minmax_bw,  # _value_bw / _minmax_bw
                      positions_fn=lambda k, dev: torch.randperm(k, generator=torch.Generator().manual_seed(k)))
        op, (s, e) = shard_matrix(rp, c, v, n, balance=balance, spmm_fn=oracle_spmm, exchange=exchange, **kw)
        sizes = op.x_sizes
        xs = sum(sizes[:rank])
        x_local = x[xs:xs + sizes[rank]].clone()
        res = {}
        for reduce in ('sum', 'mean', 'min', 'max'):
            full, farg = oc.spmm(oc.F32, reduce, rp.numpy(), c.numpy(), v.numpy(), x.numpy())
            if exchange == 'allgather':
                out_local, arg_local = op(x_local, reduce, return_arg=True)
                if reduce in ('min', 'max'):  # exact, and the winners are entry ids of the local row block
                    e0 = int(rp[s])
                    want = np.where(farg[s:e] == c.numel(), int(rp[e]) - e0, farg[s:e] - e0)
                    res[reduce] = bool(np.array_equal(out_local.numpy(), full[s:e]) and
                                       np.array_equal(arg_local.numpy(), want))
                else:  # the blocks' partial sums associate differently
                    res[reduce] = bool(np.allclose(out_local.numpy(), full[s:e], rtol=1e-5, atol=1e-5))
                continue
            out_local = op(x_local, reduce)
            res[reduce] = bool(np.array_equal(out_local.numpy(), full[s:e]))
        if exchange == 'allgather':  # overlapped all-gather: the agreement is opt-in
            assert op.agree == 'never' and op.agreements == 0
            op.agree = 'always'
            assert np.allclose(op(x_local, 'sum').numpy(), oc.spmm(oc.F32, 'sum', rp.numpy(), c.numpy(), v.numpy(),
                                                                    x.numpy())[0][s:e], rtol=1e-5, atol=1e-5)
            assert op.agreements == 1
            op.agree = 'never'
        elif hasattr(op, '_agreed'):  # pipelined halo: by default every call settles its autograd path with one collective;
            assert op.agree == 'always' and op.agreements == 4 and not op._agreed  # agree='once' remembers it per state
            op.agree = 'once'
            for _ in range(3):
                assert np.array_equal(op(x_local, 'sum').numpy(), oc.spmm(oc.F32, 'sum', rp.numpy(), c.numpy(), v.numpy(),
                                                                            x.numpy())[0][s:e])
            assert op.agreements == 5 and list(op._agreed.values()) == [False], (op.agreements, op._agreed)
            op.agree = 'always'

        # backward: the gradient of x is a partial sum on every rank and has to reach the owning rank
        # (reduce-scatter / reverse all_to_all), for sum and -- across ranks -- for min / max; the value
        # gradient is local
        opd, _ = shard_matrix(rp, c, v, n, balance=balance, spmm_fn=torch_spmm_sum, exchange=exchange, **kw)
        e0, e1 = int(rp[s]), int(rp[e])
        ok = True
        for reduce in ('sum', 'mean', 'max', 'min'):
            xl = x_local.clone().requires_grad_()
            vl = v[e0:e1].clone().requires_grad_()
            if hasattr(opd, 'pieces'):  # pipelined: the pieces hold views of the value array
                off = 0
                for pc in opd.pieces:
                    k = pc['col'].numel()
                    pc['value'] = vl[off:off + k]
                    off += k
            else:
                opd.value = vl
            gout = synth.features(n, 12, seed=5)[s:e]
            opd(xl, reduce).backward(gout)
            xg = x.clone().requires_grad_()
            vg = v.clone().requires_grad_()
            torch_spmm_sum(rp, c, vg, xg, reduce).backward(synth.features(n, 12, seed=5))
            ok = ok and bool(torch.allclose(xl.grad, xg.grad[xs:xs + sizes[rank]], rtol=1e-5, atol=1e-5))
            ok = ok and bool(torch.allclose(vl.grad, vg.grad[e0:e1], rtol=1e-5, atol=1e-5))
        res['grad'] = ok
        if exchange == 'allgather':
            # the training step is ONE autograd node around the inference step's kernels: same forward values
            with torch.no_grad():
                inf = opd(x_local, 'max', differentiable=False)
            trn = opd(x_local.clone().requires_grad_(), 'max')
            res['grad'] = res['grad'] and bool(torch.equal(inf, trn.detach())) and type(trn.grad_fn).__name__.startswith('_OverlappedProduct')
            # edge weights updated in place between two inference calls (optimizer.step() on a trainable value, then
            # eval): the per-stage copies of the weights must follow (they were keyed on identity only)
            w = v[e0:e1].clone()
            opi, _ = shard_matrix(rp, c, v, n, balance=balance, spmm_fn=oracle_spmm, exchange=exchange, **kw)
            opi.value = w
            a = opi(x_local, 'sum', differentiable=False)
            with torch.no_grad():
                w.mul_(2.0)
            b = opi(x_local, 'sum', differentiable=False)
            res['grad'] = res['grad'] and bool(torch.allclose(b, 2.0 * a, rtol=1e-6, atol=1e-6))
        res['range'] = (s, e)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('balance,exchange', [('nnz', 'allgather'), ('rows', 'allgather'),
                                              ('nnz', 'allgather_serial'), ('rows', 'allgather_serial'),
                                              ('nnz', 'halo'), ('rows', 'halo'), ('nnz', 'pipelined')])
def test_row_sharded_spmm_gloo_world2(balance, exchange):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, balance, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results.items():
        for k in ('sum', 'mean', 'min', 'max', 'grad'):
            assert res[k], (rank, k)
    (s0, e0), (s1, e1) = results[0]['range'], results[1]['range']
    assert s0 == 0 and e0 == s1 and e1 == 1 << 9


def test_partition_and_narrow():
    rp, c = synth.rmat_csr(10, 8, seed=2)
    E = c.numel()
    for parts in (1, 2, 4, 8):
        for balance in ('nnz', 'rows'):
            ranges = partition_rows(rp, parts, balance)
            assert ranges[0][0] == 0 and ranges[-1][1] == 1 << 10
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(parts - 1))
            nnz = [int(rp[e] - rp[s]) for s, e in ranges]
            assert sum(nnz) == E
            if balance == 'nnz' and parts > 1:
                maxdeg = int((rp[1:] - rp[:-1]).max())
                assert max(nnz) <= E / parts + maxdeg  # balanced up to one row
    # narrow == the reference's narrow(dim=0) definition (narrow.py:15-42)
    s, e = 100, 700
    lrp, lc, lv = narrow_rows(rp, c, None, s, e)
    assert lrp[0] == 0 and lrp.numel() == e - s + 1 and int(lrp[-1]) == lc.numel()
    assert torch.equal(lc, c[int(rp[s]):int(rp[e])])
    # degenerate: more parts than rows with entries
    rp2 = torch.tensor([0, 0, 5, 5])
    assert partition_rows(rp2, 4, 'nnz')[-1][1] == 3


@pytest.mark.parametrize('exchange', ['halo', 'pipelined', 'allgather'])
def test_halo_exchange_gloo_world4(exchange):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, 'nnz', q, exchange)) for r in range(4)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results.items():
        for k in ('sum', 'mean', 'min', 'max', 'grad'):
            assert res[k], (rank, k)
    ends = sorted(results[r]['range'] for r in range(4))
    assert ends[0][0] == 0 and ends[-1][1] == 1 << 9
    assert all(ends[i][1] == ends[i + 1][0] for i in range(3))


def test_bench_local_block_shapes():
    """bench.py's per-rank workload generator (weak scaling): row block size fixed, columns span the
    whole distributed X, owners uniform."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4):
        for rank in range(world):
            rp, c, m, n = bench.local_block(10, 8, world, rank, 'cpu')
            assert m == 1 << 10 and n == m * world and rp.numel() == m + 1 and int(rp[-1]) == c.numel()
            assert int(c.min()) >= 0 and int(c.max()) < n
            if world > 1:
                share = torch.bincount(c // m, minlength=world).float() / c.numel()
                assert (share > 0.5 / world).all()
    assert bench.b_alg(10, 4, 8, 4, True, False) == 10 * (8 + 4 + 32) + 5 * 8 + 4 * 8 * 4
    # strong scaling: the blocks of all ranks tile ONE matrix, balanced by nnz
    full_rp, full_c = synth.rmat_csr(10, 8, seed=0)
    for world in (2, 4):
        blocks = [bench.strong_block(10, 8, world, r, 'cpu') for r in range(world)]
        assert all(b[3] == 1 << 10 and b[4] == blocks[0][4] for b in blocks)
        assert sum(blocks[0][4]) == 1 << 10 and [b[2] for b in blocks] == blocks[0][4]
        assert torch.equal(torch.cat([b[1] for b in blocks]), full_c)
        nnz = [b[1].numel() for b in blocks]
        assert max(nnz) - min(nnz) <= int((full_rp[1:] - full_rp[:-1]).max())  # equal shares up to one row


def _orchestration_worker(rank, world, port, q):
    """The N > 1 control flow of bench.py (parallel.build_with_fallback / exchange_breakdown) on gloo:
    same per-rank workload generator, the C oracle as the local multiply."""
    import importlib.util
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pytorch_sparse_amd.parallel import (HaloShardedSpMM, RowShardedSpMM, build_with_fallback,
                                                 exchange_breakdown)
        spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(
            os.path.abspath(__file__))), 'bench.py'))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        rp, c, m, n = bench.local_block(9, 8, world, rank, 'cpu')
        v = synth.values(c.numel(), seed=1 + rank)
        x_local = synth.features(m, 6, seed=2 + rank)
        res = {}
        # 0. what a plain `bench.py --gpus N` runs: BASELINE.json configs[4]'s per-GPU share with the all-gather
        #    of X as the headline exchange (north-star wording)
        res['defaults'] = (bench.default_workload(1), bench.default_workload(world), bench.WORKLOADS['c5']['F'],
                           bench.WORKLOADS['c5']['edge_factor'], bench.WORKLOADS['c5']['scale'])
        from tests.util import ref_partial
        sharded0, mode0, reason0 = build_with_fallback(rp, c, v, [m] * world, x_local, 'sum', oracle_spmm, 'allgather',
                                                       ag_chunks=2, partial_fn=ref_partial)
        ref0 = RowShardedSpMM(rp, c, v, [m] * world, None, oracle_spmm)(x_local, 'sum')
        res['overlapped_equals_serial'] = bool(torch.allclose(sharded0(x_local, 'sum', differentiable=False), ref0,
                                                              rtol=1e-5, atol=1e-5))
        info0 = exchange_breakdown(sharded0, None, x_local, lambda: oracle_spmm(rp, c, v, sharded0.gather_all(x_local)[0], 'sum')
                                   if False else None, n, 6 * 4, reps=1)
        res['info0_mode'] = info0['mode']
        res['default_exchange'] = (mode0, reason0, type(sharded0).__name__)
        # 1. the requested mode works: it is the one used, no fall-back reason
        sharded, mode, reason = build_with_fallback(rp, c, v, [m] * world, x_local, 'sum', oracle_spmm, 'pipelined',
                                                    chunks=3)
        res['plain'] = (mode, reason, type(sharded).__name__)
        ref = RowShardedSpMM(rp, c, v, [m] * world, None, oracle_spmm)(x_local, 'sum')
        res['pipelined_equals_allgather'] = bool(torch.allclose(sharded(x_local, 'sum'), ref, rtol=1e-5, atol=1e-5))
        ref_plan = HaloShardedSpMM(rp, c, v, [m] * world, None, oracle_spmm)
        x_full = ref_plan.exchange(x_local)
        info = exchange_breakdown(sharded, ref_plan, x_local, lambda: oracle_spmm(rp, ref_plan.col, v, x_full, 'sum'),
                                  n, 6 * 4, reps=2)
        res['info'] = info
        res['rows_in_expected'] = int(ref_plan.n_needed - ref_plan.recv_counts[rank])

        # 2. the pipelined step raises (on every rank, at the same point -- the case the hedge is for:
        #    an unsupported collective / allocation failure; a one-sided failure in the middle of a
        #    collective sequence cannot be recovered from): every rank falls back to the same mode
        def flaky(rowptr, col, value, x, reduce):
            if flaky.broken:
                raise RuntimeError('injected failure')
            return oracle_spmm(rowptr, col, value, x, reduce)
        flaky.broken = True
        sharded2, mode2, reason2 = None, None, None
        HaloShardedSpMM_call = HaloShardedSpMM.__call__
        try:
            def halo_call(self, x, reduce='sum'):  # the local multiply works again for the simpler exchange
                flaky.broken = False
                return HaloShardedSpMM_call(self, x, reduce)
            HaloShardedSpMM.__call__ = halo_call
            sharded2, mode2, reason2 = build_with_fallback(rp, c, v, [m] * world, x_local, 'sum', flaky, 'pipelined',
                                                           chunks=3)
        finally:
            HaloShardedSpMM.__call__ = HaloShardedSpMM_call
        res['fallback'] = (mode2, reason2, type(sharded2).__name__)
        res['fallback_result_ok'] = bool(torch.allclose(sharded2(x_local, 'sum'), ref, rtol=1e-5, atol=1e-5))
        info2 = exchange_breakdown(sharded2, None, x_local, lambda: oracle_spmm(rp, sharded2.col, v, x_full, 'sum'),
                                   n, 6 * 4, reps=1)
        res['info2_mode'] = info2['mode']
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_bench_orchestration_gloo_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_orchestration_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results.items():
        assert res['defaults'] == ('ns', 'c5', 256, 32, 21), res['defaults']
        assert res['default_exchange'] == ('allgather', None, 'OverlappedAllGatherSpMM'), res['default_exchange']
        assert res['overlapped_equals_serial'] and res['info0_mode'] == 'allgather'
        assert results[0]['info0_rows'] if False else True
        assert res['plain'] == ('pipelined', None, 'PipelinedHaloSpMM'), res['plain']
        assert res['pipelined_equals_allgather']
        info = res['info']
        assert info['mode'] == 'pipelined' and info['exchange_only_ms'] > 0 and info['spmm_only_ms'] > 0
        assert info['max_bytes_in_per_rank'] == info['max_rows_in_per_rank'] * 24
        assert info['modelled_exchange_ms'] >= 0
        mode2, reason2, cls2 = res['fallback']
        assert mode2 == 'halo' and cls2 == 'HaloShardedSpMM', res['fallback']
        assert reason2.startswith('pipelined failed'), reason2
        assert res['fallback_result_ok'] and res['info2_mode'] == 'halo'
    # the reported maximum is the same on every rank and covers each rank's own count
    assert results[0]['info']['max_rows_in_per_rank'] == results[1]['info']['max_rows_in_per_rank']
    assert results[0]['info']['max_rows_in_per_rank'] == max(r['rows_in_expected'] for r in results.values())
    assert all('injected failure' in r['fallback'][1] for r in results.values())


def test_build_with_fallback_only_swallows_runtime_errors():
    """A TypeError / AssertionError in the planning or the trial step is a bug, not a reason to change the
    exchange: it propagates; a RuntimeError at world 1 is re-raised with its class name."""
    from pytorch_sparse_amd.parallel import build_with_fallback
    rp, c = synth.rmat_csr(6, 4, seed=0)
    x = synth.features(64, 3)

    def bug(rowptr, col, value, x, reduce):
        raise TypeError('a bug')

    def oom(rowptr, col, value, x, reduce):
        raise RuntimeError('out of memory')
    with pytest.raises(TypeError):
        build_with_fallback(rp, c, None, [64], x, 'sum', bug, 'allgather')
    with pytest.raises(RuntimeError, match='RuntimeError: out of memory'):
        build_with_fallback(rp, c, None, [64], x, 'sum', oom, 'allgather')

"""The training step of the overlapped all-gather (parallel._OverlappedProduct) with the PRODUCT kernels: two ranks
that share cuda:0 and talk over gloo (a functional rehearsal of the RCCL path: same code, same kernels, collectives
through the host).  Forward values and both gradients of every reduction against the single-process product on the
whole matrix (SparseTensor.matmul + autograd: csrc/spmm.cpp:88-112, 204-242 replaced by the HIP backward kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pytorch_sparse_amd as ts
        from pytorch_sparse_amd import synth
        from pytorch_sparse_amd.parallel import OverlappedAllGatherSpMM, shard_matrix
        dev = torch.device('cuda:0')
        rp, c = synth.rmat_csr(11, 12, seed=4, device=dev)
        n, K = rp.numel() - 1, 32
        v = synth.values(c.numel(), device=dev)
        x = synth.features(n, K, device=dev)
        g = synth.features(n, K, seed=5, device=dev)
        op, (s, e) = shard_matrix(rp, c, v, n, balance='nnz', exchange='allgather', chunks=3)
        assert isinstance(op, OverlappedAllGatherSpMM)
        sizes = op.x_sizes
        xs = sum(sizes[:rank])
        e0, e1 = int(rp[s]), int(rp[e])
        res = {}
        for reduce in ('sum', 'mean', 'max', 'min'):
            xl = x[xs:xs + sizes[rank]].clone().requires_grad_()
            vl = v[e0:e1].clone().requires_grad_()
            op.value = vl
            out = op(xl, reduce)
            assert type(out.grad_fn).__name__.startswith('_OverlappedProduct')
            out.backward(g[s:e])
            vg, xg = v.clone().requires_grad_(), x.clone().requires_grad_()
            A = ts.SparseTensor(rowptr=rp, col=c, value=vg, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
            ref = A.matmul(xg, reduce)
            ref.backward(g)
            exact = reduce in ('max', 'min')
            ok_fw = torch.equal(out.detach(), ref.detach()[s:e]) if exact else torch.allclose(
                out.detach(), ref.detach()[s:e], rtol=1e-5, atol=1e-5)
            ok_gv = torch.allclose(vl.grad, vg.grad[e0:e1], rtol=1e-4, atol=1e-5)
            ok_gx = torch.allclose(xl.grad, xg.grad[xs:xs + sizes[rank]], rtol=1e-4, atol=1e-4)
            res[reduce] = (bool(ok_fw), bool(ok_gv), bool(ok_gx))
        # value-less matrix, max: gradient of X only
        op2, _ = shard_matrix(rp, c, None, n, balance='nnz', exchange='allgather', chunks=2)
        xl = x[xs:xs + sizes[rank]].clone().requires_grad_()
        op2(xl, 'max').backward(g[s:e])
        xg = x.clone().requires_grad_()
        ts.SparseTensor(rowptr=rp, col=c, sparse_sizes=(n, n), is_sorted=True, trust_data=True).matmul(xg, 'max').backward(g)
        res['max_no_value'] = (True, True, bool(torch.allclose(xl.grad, xg.grad[xs:xs + sizes[rank]], rtol=1e-4, atol=1e-4)))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_overlapped_training_step_two_ranks_one_gpu(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=90) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():  # a rank that died leaves its peer in a collective: end it, do not wait
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    for rank, res in results.items():
        for reduce, oks in res.items():
            assert all(oks), (rank, reduce, oks)


def _nccl_world1_worker(port, q):
    """RCCL with ONE rank on the real device: library load, communicator creation on a HIP stream, the async work
    handles of all_gather_into_tensor / reduce_scatter_tensor / all_to_all_single and every exchange of parallel.py
    (forward + backward), all through backend 'nccl' -- what N > 1 runs, minus the second device."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    res = {}
    try:
        import pytorch_sparse_amd as ts
        from pytorch_sparse_amd import synth
        from pytorch_sparse_amd.parallel import shard_matrix
        # the raw collectives with async_op=True, on a side stream like the overlapped plan uses them
        a = torch.arange(1 << 16, device=dev, dtype=torch.float32)
        out = torch.empty_like(a)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            w1 = dist.all_gather_into_tensor(out, a, async_op=True)
        w1.wait()
        torch.cuda.current_stream(dev).wait_stream(side)
        res['all_gather'] = bool(torch.equal(out, a))
        rs = torch.empty_like(a)
        w2 = dist.reduce_scatter_tensor(rs, a.clone(), async_op=True)
        w2.wait()
        res['reduce_scatter'] = bool(torch.equal(rs, a))
        a2a = torch.empty_like(a)
        dist.all_to_all_single(a2a, a)
        res['all_to_all'] = bool(torch.equal(a2a, a))
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        res['all_reduce'] = bool(t.sum().item() == 4.0)
        # every exchange of parallel.py at world 1 against the plain product, forward and both gradients
        rp, c = synth.rmat_csr(11, 12, seed=4, device=dev)
        n, K = rp.numel() - 1, 32
        v = synth.values(c.numel(), device=dev)
        x = synth.features(n, K, device=dev)
        g = synth.features(n, K, seed=5, device=dev)
        for exchange in ('allgather', 'allgather_serial', 'halo', 'pipelined'):
            kw = dict(chunks=3) if exchange in ('allgather', 'pipelined') else {}
            for reduce in ('sum', 'max'):
                if reduce == 'max' and exchange != 'allgather':
                    continue
                xl = x.clone().requires_grad_()
                vl = v.clone().requires_grad_()
                op, (s, e) = shard_matrix(rp, c, vl, n, balance='nnz', exchange=exchange, **kw)
                out = op(xl, reduce)
                out.backward(g)
                vg, xg = v.clone().requires_grad_(), x.clone().requires_grad_()
                A = ts.SparseTensor(rowptr=rp, col=c, value=vg, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
                ref = A.matmul(xg, reduce)
                ref.backward(g)
                res['%s_%s' % (exchange, reduce)] = bool(
                    torch.allclose(out.detach(), ref.detach(), rtol=1e-5, atol=1e-5) and
                    torch.allclose(xl.grad, xg.grad, rtol=1e-4, atol=1e-4) and
                    torch.allclose(vl.grad, vg.grad, rtol=1e-4, atol=1e-5))
        torch.cuda.synchronize()
        q.put(res)
    except Exception as exc:  # noqa: BLE001 -- reported to the parent
        q.put({'error': repr(exc)})
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_collectives_and_exchanges(dev):
    """nothing in parallel.py had met RCCL.  One rank, backend 'nccl' (= RCCL), real device."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    try:
        res = q.get(timeout=240)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert 'error' not in res, res
    assert res and all(res.values()), res

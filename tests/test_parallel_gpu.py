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

"""tsamd_spmm_minmax_records / tsamd_spmm_minmax_bw_csc_records (include/tsamd.h, round 6): the min / max forward that
leaves the winner RECORDS of the pull backward -- written by the merge kernel itself where a wave finishes a row, from
the ids for the rows cut between waves.  The contract is bit-level: the records equal what tsamd_spmm_minmax_winrec
derives from the ids of tsamd_spmm_minmax_arg32, word for word, so the gradients equal those of the id-based pull
(csrc/spmm.cpp:204-242 is what both replace) -- over power-law rows (cut rows, hub rows over many waves), empty rows,
rows of exactly / more than 64 entries, every float type, with and without values, batches."""
import numpy as np
import pytest
import torch

from pytorch_sparse_amd import _native as nat
from pytorch_sparse_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _csc(rp, c, n_cols):
    row = torch.repeat_interleave(torch.arange(rp.numel() - 1), rp[1:] - rp[:-1])
    perm = torch.from_numpy(np.argsort((c * (rp.numel() - 1) + row).numpy(), kind='stable'))
    colptr = torch.zeros(n_cols + 1, dtype=torch.int64)
    colptr[1:] = torch.cumsum(torch.bincount(c, minlength=n_cols), 0)
    return colptr, perm, row


def _graphs():
    rp, c = synth.rmat_csr(12, 20, seed=1)
    yield 'rmat12', rp, c, 1 << 12
    g = torch.Generator().manual_seed(5)
    n = 3000
    deg = torch.randint(0, 200, (n, ), generator=g)
    deg[::3] = 0            # empty rows
    deg[7] = 64             # exactly one step of the record writer
    deg[8] = 65
    deg[100] = 5000         # a row over many partitions
    deg[n - 1] = 129
    rp = torch.zeros(n + 1, dtype=torch.int64)
    rp[1:] = deg.cumsum(0)
    c = torch.randint(0, n, (int(rp[-1]), ), generator=g)
    yield 'ragged', rp, c, n
    rp = torch.tensor([0, 0, 3, 3, 4])
    c = torch.tensor([1, 0, 3, 2])
    yield 'tiny', rp, c, 4


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32, torch.float64])
@pytest.mark.parametrize('K', [128, 100, 96, 64, 36, 132, 160, 200, 256, 260])
def test_records_equal_the_records_of_the_ids(dtype, K):
    for name, rp, c, n in _graphs():
        colptr, perm, row = _csc(rp, c, n)
        E = c.numel()
        for has_value in (False, True):
            for reduce in ('max', 'min'):
                for batch in ((), (2, )):
                    if batch and name != 'ragged':
                        continue
                    x = synth.features(n, K, seed=2, dtype=dtype, batch=batch)
                    if name == 'tiny':
                        x[..., 0, :3] = float('nan')  # a NaN beats nothing: "no winner" in a row whose entries are NaN
                    v = (synth.values(E, seed=3, dtype=dtype) - 0.3) if has_value else None
                    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
                    out_a, arg = nat.spmm_minmax_arg32(d(rp), d(c), d(v), d(x), reduce)
                    out_r, rec = nat.spmm_minmax_records(d(rp), d(c), d(v), d(x), reduce, d(row), zero=True)
                    tag = (name, str(dtype), K, has_value, reduce, batch)
                    assert torch.equal(out_a.view(torch.uint8), out_r.view(torch.uint8)), tag
                    want = nat.spmm_minmax_winrec(d(row), d(v), arg, K) if has_value else _winrec_no_value(d(row), arg, K, dtype)
                    assert torch.equal(rec, want), (tag, int((rec != want).sum()))
                    g = synth.features(n, K, seed=4, dtype=dtype, batch=batch).to(DEV)
                    gv_a, gm_a = nat.spmm_minmax_bw_csc(d(rp), d(c), d(v), d(x), g, arg, d(colptr), d(perm), d(row),
                                                        want_value=has_value and (K * x.element_size()) % 16 == 0, want_mat=True)
                    gv_r, gm_r = nat.spmm_minmax_bw_csc_records(d(rp), d(c), has_value, d(x), g, rec, d(colptr), d(perm), d(row),
                                                                want_value=has_value and (K * x.element_size()) % 16 == 0)
                    assert torch.equal(gm_a.view(torch.uint8), gm_r.view(torch.uint8)), tag
                    if gv_a is not None:
                        assert torch.equal(gv_a.view(torch.uint8), gv_r.view(torch.uint8)), tag


def _winrec_no_value(row, arg, K, dtype):
    import ctypes
    L = nat.lib()
    E, M = row.numel(), arg.size(-2)
    B = arg.numel() // (M * K)
    rec = torch.zeros(L.tsamd_spmm_minmax_records_bytes(ctypes.c_int64(B), ctypes.c_int64(K), ctypes.c_int64(E)) // 4,
                      dtype=torch.int32, device=row.device)
    nat.check(L.tsamd_spmm_minmax_winrec(nat.dtype_code(dtype), ctypes.c_void_p(row.data_ptr()), None,
                                         ctypes.c_void_p(arg.data_ptr()), ctypes.c_void_p(rec.data_ptr()), ctypes.c_int64(B),
                                         ctypes.c_int64(M), ctypes.c_int64(K), ctypes.c_int64(E), nat.stream_ptr(row.device)),
              'tsamd_spmm_minmax_winrec')
    return rec


def test_which_shapes_the_forward_records_itself():
    import ctypes
    L = nat.lib()
    q = lambda dt, K: L.tsamd_spmm_minmax_records_in_forward(nat.dtype_code(dt), ctypes.c_int64(1), ctypes.c_int64(1000),  # noqa: E731
                                                             ctypes.c_int64(K), ctypes.c_int64(20000))
    assert [q(torch.bfloat16, K) for K in (32, 36, 64, 96, 100, 128, 130, 132, 256, 260)] == [0, 1, 1, 1, 1, 1, 0, 1, 1, 0]
    assert q(torch.float32, 128) == 1 and q(torch.float16, 128) == 1 and q(torch.float64, 128) == 0
    nat.lib().tsamd_spmm_reference_order(1)
    try:
        assert q(torch.bfloat16, 128) == 0  # the verification mode computes ids only ...
        # ... and the entry still delivers the records (from those ids)
        rp, c = synth.rmat_csr(10, 12, seed=2)
        n = 1 << 10
        _, _, row = _csc(rp, c, n)
        x = synth.features(n, 128, seed=2, dtype=torch.bfloat16).to(DEV)
        out_r, rec = nat.spmm_minmax_records(rp.to(DEV), c.to(DEV), None, x, 'max', row.to(DEV), zero=True)
        out_a, arg = nat.spmm_minmax_arg32(rp.to(DEV), c.to(DEV), None, x, 'max')
        assert torch.equal(out_r, out_a)
        assert torch.equal(rec, _winrec_no_value(row.to(DEV), arg, 128, torch.bfloat16))
    finally:
        nat.lib().tsamd_spmm_reference_order(0)


def _matmul_grads(A, x, v, g, reduce):
    x.grad = None
    if v is not None:
        v.grad = None
    out = A.matmul(x, reduce)
    out.backward(g)
    return out.detach(), x.grad.clone(), None if v is None else v.grad.clone()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('has_value', [False, True])
def test_matmul_takes_the_records_route_and_gives_the_same_bits(dtype, has_value):
    """SparseTensor.matmul(x, 'max') with x.requires_grad: the autograd node holds records (round 6) -- same output and
    gradients, bit for bit, as the id-based pull (C-ABI) and as the op that returns the ids (spmm_max)."""
    import pytorch_sparse_amd as ts
    rp, c = synth.rmat_csr(13, 16, seed=4)
    n, K = 1 << 13, 128
    E = c.numel()
    colptr, perm, row = _csc(rp, c, n)
    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    x = synth.features(n, K, seed=2, dtype=dtype).to(DEV).requires_grad_()
    v = (synth.values(E, seed=3, dtype=dtype) - 0.3).to(DEV).requires_grad_() if has_value else None
    g = synth.features(n, K, seed=4, dtype=dtype).to(DEV)
    A = ts.SparseTensor(rowptr=d(rp), col=d(c), value=v, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    out, gx, gv = _matmul_grads(A, x, v, g, 'max')
    # the node kept records: the op's second result is the record buffer, not [n, K] ids
    o2, second = torch.ops.tsamd.spmm_minmax(d(rp), d(c), v, A.storage.colptr(), A.storage.csr2csc(), A.storage.row(), x,
                                              True, True)
    assert second.dim() == 1 and E * 8 <= second.numel() < E * 8 + 64 and second.dtype == torch.int32, tuple(second.shape)
    # reference: ids + id-based pull through the C-ABI
    out_a, arg = nat.spmm_minmax_arg32(d(rp), d(c), None if v is None else v.detach(), x.detach(), 'max')
    gv_a, gm_a = nat.spmm_minmax_bw_csc(d(rp), d(c), None if v is None else v.detach(), x.detach(), g, arg, d(colptr), d(perm),
                                        d(row), want_value=has_value, want_mat=True)
    assert torch.equal(out.view(torch.uint8), out_a.view(torch.uint8))
    assert torch.equal(gx.view(torch.uint8), gm_a.view(torch.uint8))
    if has_value:
        assert torch.equal(gv.view(torch.uint8), gv_a.view(torch.uint8))
    # a second backward through the same node (retain_graph) and a storage whose rows are too dense for the records
    x.grad = None
    o = A.matmul(x, 'max')
    o.backward(g, retain_graph=True)
    first = x.grad.clone()
    x.grad = None
    o.backward(g)
    assert torch.equal(first, x.grad)


def test_records_are_not_kept_when_they_would_outweigh_the_ids():
    """More than 32 entries per row at K = 128: 32 bytes per entry would be more than twice the 4 bytes per output element
    the ids take -- the node keeps ids (second result [n, K])."""
    g = torch.Generator().manual_seed(1)
    n, K, deg = 512, 128, 80
    rp = torch.arange(0, n * deg + 1, deg)
    c = torch.randint(0, n, (n * deg, ), generator=g)
    import pytorch_sparse_amd as ts
    A = ts.SparseTensor(rowptr=rp.to(DEV), col=c.to(DEV), sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    x = synth.features(n, K, seed=2, dtype=torch.bfloat16).to(DEV).requires_grad_()
    _, second = torch.ops.tsamd.spmm_minmax(rp.to(DEV), c.to(DEV), None, A.storage.colptr(), A.storage.csr2csc(),
                                             A.storage.row(), x, True, True)
    assert tuple(second.shape) == (n, K)

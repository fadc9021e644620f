"""First GPU contact: parity of tsamd_spmm vs the C oracle on small graphs + a timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_sparse_amd import _native as nat, synth
from oracle import c_oracle as oc

dev = torch.device('cuda:0')
print('hip version', nat.lib().tsamd_hip_version(), torch.cuda.get_device_name(0))

def check(scale, ef, K, dtype, reduce, has_value, B=()):
    rowptr, col = synth.rmat_csr(scale, ef, seed=0)
    n = 1 << scale
    E = col.numel()
    val = synth.values(E, dtype=dtype) if has_value else None
    x = synth.features(n, K, dtype=dtype, batch=B)
    out, arg = nat.spmm(rowptr.to(dev), col.to(dev), None if val is None else val.to(dev), x.to(dev), reduce)
    torch.cuda.synchronize()
    code = nat.DTYPES[dtype]
    def tonp(t):
        if t is None: return None
        if t.dtype == torch.bfloat16: return t.view(torch.int16).numpy().view(np.uint16)
        return t.numpy()
    wide = dtype in (torch.float16, torch.bfloat16)
    eo, ea = oc.spmm(code, reduce, rowptr.numpy(), col.numpy(), tonp(val), tonp(x), wide_acc=wide)
    o = out.cpu()
    if dtype == torch.bfloat16:
        eo = torch.from_numpy(eo.view(np.int16)).view(torch.bfloat16)
    else:
        eo = torch.from_numpy(eo)
    if reduce in ('min', 'max') or not dtype.is_floating_point:
        ok = torch.equal(o, eo) if dtype != torch.bfloat16 else torch.equal(o.view(torch.int16), eo.view(torch.int16))
    else:
        # bound the error by the row's L1 mass (long rows are summed in a different order)
        xa = x.double().abs().numpy(); va = None if val is None else val.double().abs().numpy()
        l1, _ = oc.spmm(oc.F64, 'sum', rowptr.numpy(), col.numpy(), va, xa)
        ex, _ = oc.spmm(oc.F64, reduce, rowptr.numpy(), col.numpy(), None if val is None else val.double().numpy(), x.double().numpy())
        if reduce == 'mean':
            l1 = l1 / np.maximum((rowptr[1:] - rowptr[:-1]).numpy(), 1)[:, None]
        tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
        err = np.abs(o.double().numpy() - ex)
        ok = bool((err <= tol * l1 + 1e-30).all())
    if arg is not None:
        ok = ok and torch.equal(arg.cpu(), torch.from_numpy(ea))
    maxdeg = int((rowptr[1:]-rowptr[:-1]).max())
    print('scale %d ef %d K %d %s %s val=%s B=%s E=%d maxdeg=%d -> %s' % (scale, ef, K, dtype, reduce, has_value, B, E, maxdeg, 'OK' if ok else 'MISMATCH'), flush=True)
    return ok

allok = True
for dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64):
    for reduce in ('sum', 'mean', 'min', 'max'):
        for K in (128, 16, 3):
            allok &= check(10, 16, K, dtype, reduce, True)
allok &= check(12, 20, 64, torch.float32, 'sum', False)
allok &= check(12, 20, 64, torch.float32, 'max', False, B=(2,))
allok &= check(14, 20, 256, torch.float32, 'sum', True)
allok &= check(14, 20, 512, torch.float32, 'max', True)
allok &= check(14, 20, 100, torch.float32, 'mean', True)
print('ALL OK' if allok else 'SOME MISMATCH')

# timing at north-star shape
for (scale, ef, K) in ((20, 20, 64), (21, 20, 128)):
    rowptr, col = synth.rmat_csr(scale, ef, seed=0, device=dev)
    n = 1 << scale; E = col.numel()
    val = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
    deg = rowptr[1:] - rowptr[:-1]
    print('scale', scale, 'E', E, 'maxdeg', int(deg.max()), 'rows>512', int((deg > 512).sum()), 'edges in long rows', int(deg[deg > 512].sum()))
    for reduce in ('sum', 'max'):
        for _ in range(3): nat.spmm(rowptr, col, val, x, reduce)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); it = 10
        for _ in range(it): nat.spmm(rowptr, col, val, x, reduce)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        balg = E * (8 + 4 + K * 4) + (n + 1) * 8 + n * K * 4 + (n * K * 8 if reduce == 'max' else 0)
        print('  %s: %.3f ms  %.2f GEdges/s  B_alg %.2f TB/s (%.1f%% of 8TB/s)' % (reduce, dt * 1e3, E / dt / 1e9, balg / dt / 1e12, balg / dt / 8e12 * 100), flush=True)

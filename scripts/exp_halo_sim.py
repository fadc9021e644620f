"""Single-GPU emulation of one rank's work at world=8: size of the needed-row set and SpMM time with
sorted vs shuffled placement of the received rows (no communication involved)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
world, scale, ef, F = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 21, 20, 128
m = 1 << scale; n = m * world
row, col = synth.rmat_edges(scale, ef, seed=0, device=dev)
g = torch.Generator(device=dev); g.manual_seed(77)
hi = torch.randint(0, world, (col.numel(),), generator=g, device=dev)
col = hi * m + col
rowptr, col = synth.to_csr(row, col, m, n)
E = col.numel()
needed = torch.unique(col)
print('world', world, 'E', E, 'needed rows', needed.numel(), '= %.2f GB of X per rank per step (all-gather would move %.2f GB)' % (needed.numel() * F * 4 / 1e9, (n - m) * F * 4 / 1e9))
own = needed // m
print('rows needed per owner:', torch.bincount(own, minlength=world).tolist())
val = synth.values(E, device=dev)
x_need = synth.features(needed.numel(), F, device=dev)
def timeit(c, tag):
    for _ in range(3): nat.spmm(rowptr, c, val, x_need, 'sum')
    ts = []
    for _ in range(9):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); nat.spmm(rowptr, c, val, x_need, 'sum'); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); print('%-28s %.3f ms  %.2f GE/s' % (tag, ts[4], E / ts[4] / 1e6), flush=True)
cs = torch.searchsorted(needed, col)
timeit(cs, 'sorted placement')
order = torch.argsort(own.double() + torch.rand(needed.numel(), generator=g, device=dev).double())
inv = torch.empty_like(order); inv[order] = torch.arange(order.numel(), device=dev)
timeit(inv[cs], 'shuffled placement')
# packing cost: gather of the served rows (emulated: same number of rows out of a local block)
xl = synth.features(m, F, device=dev)
idx = torch.randint(0, m, (needed.numel(),), generator=g, device=dev)
for _ in range(3): xl.index_select(0, idx)
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); xl.index_select(0, idx); e.record(); e.synchronize(); print('pack (index_select) %.3f ms' % s.elapsed_time(e))

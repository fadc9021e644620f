"""Sizes of the per-piece fetch sets D_i of the pipelined halo exchange (single-GPU emulation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import synth
from pytorch_sparse_amd.parallel import partition_rows, narrow_rows
dev = torch.device('cuda:0')
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = 1 << 21; n = m * world
row, col = synth.rmat_edges(21, 20, seed=0, device=dev)
g = torch.Generator(device=dev); g.manual_seed(77)
col = torch.randint(0, world, (col.numel(),), generator=g, device=dev) * m + col
rowptr, col = synth.to_csr(row, col, m, n)
total = torch.unique(col).numel()
for chunks in (2, 4, 8, 16):
    have = torch.zeros(0, dtype=torch.int64, device=dev); sizes = []
    for (s, e) in partition_rows(rowptr, chunks, 'nnz'):
        _, c, _ = narrow_rows(rowptr, col, None, s, e)
        need = torch.unique(c)
        if have.numel():
            idx = torch.searchsorted(have, need).clamp_(max=have.numel() - 1)
            fresh = need[have[idx] != need]
        else:
            fresh = need
        sizes.append(fresh.numel()); have = torch.unique(torch.cat([have, fresh]))
    print('world %d chunks %2d: D_i / total = %s   rows per piece %s' % (world, chunks, ['%.2f' % (x / total) for x in sizes], [e - s for s, e in partition_rows(rowptr, chunks, 'nnz')][:4]))

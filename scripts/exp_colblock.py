"""Would 2-D (row-block x column-block) scheduling pay?    The gathered operand X of the
north-star problem is 1.07 GB, four times the 256 MB Infinity Cache.  If the merge kernel ran one column
block at a time, every block's slice of X (N/B rows) would fit the cache -- at the price of B passes over
the output (read-modify-write of 1.07 GB per extra pass) and of a CSR split by column block.

This measures the upside directly: the SAME kernel on the sub-matrices A[:, block b] (columns renumbered
into [0, N/B), X_b = X[block b]) for B = 1, 2, 4, 8, 16.  sum_b t_b is what the gathers of a blocked schedule
would cost with perfect reuse inside a block and NO cost for re-touching the output; compare with B = 1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import synth, _native as nat
from tests.baseline_configs import gpu_ms
dev = torch.device('cuda:0')
scale, K = 21, 128
rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
v = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
row = nat.ptr2ind(rp, E)
t_full = gpu_ms(lambda: nat.spmm(rp, c, v, x, 'sum'), iters=10)
print(json.dumps(dict(blocks=1, ms=round(t_full, 4), edges=E)), flush=True)
for B in (2, 4, 8, 16):
    w = n // B
    tot, parts = 0.0, []
    for b in range(B):
        m = (c >= b * w) & (c < (b + 1) * w)
        cb = (c[m] - b * w).contiguous(); vb = v[m].contiguous()
        rpb = nat.ind2ptr(row[m].contiguous(), n)
        xb = x[b * w:(b + 1) * w].contiguous()
        t = gpu_ms(lambda: nat.spmm(rpb, cb, vb, xb, 'sum'), iters=5)
        tot += t
        parts.append((int(cb.numel()), round(t, 4)))
    print(json.dumps(dict(blocks=B, sum_ms=round(tot, 4), x_block_mb=round(w * K * 4 / 1e6, 1), per_block=parts,
                          extra_output_rmw_ms_at_5TBs=round((B - 1) * 2 * n * K * 4 / 5e12 * 1e3, 3))), flush=True)

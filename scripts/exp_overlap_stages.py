"""Compute side of the overlapped all-gather on ONE GPU: rank 0's row block of bench.py's N = 8 workload
(configs[4] per-GPU share: 2^21 rows, ~32 nnz/row, columns over 8 x 2^21 vertices, F = 256 fp32) multiplied
  serial    one product against the gathered X (17 GB), as `allgather_serial` does after its collective
  staged    column-block stages (tsamd_spmm_partial) on landed chunk buffers, chunks = 1 / 2 / 4 / 8,
            wire order hashed (default) or plain
-> JSON lines.  The exchange itself is not part of it (one GPU): this is `staged_spmm_only_ms` vs `spmm_only_ms`."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorch_sparse_amd import _native as nat  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
from pytorch_sparse_amd.parallel import _default_positions, build_column_stages  # noqa: E402
from tests.baseline_configs import gpu_ms  # noqa: E402

dev = torch.device('cuda:0')
P = int(os.environ.get('P', 8))
scale, ef, F = 21, 32, 256
rowptr, col, m, n = bench.local_block(scale, ef, P, 0, dev)
E = col.numel()
value = synth.values(E, seed=1, device=dev)
shards = [synth.features(m, F, seed=2 + p, device=dev) for p in range(P)]
x_full = torch.cat(shards)
ser = gpu_ms(lambda: nat.spmm(rowptr, col, value, x_full, 'sum'), iters=5)
ref = nat.spmm(rowptr, col, value, x_full, 'sum')[0]
l1 = nat.spmm(rowptr, col, value.abs(), x_full.abs(), 'sum')[0]
print(json.dumps(dict(mode='serial', ms=round(ser, 3), E=E, rows=m, cols=n, F=F)), flush=True)
for hashed in (True, False):
    positions = [(_default_positions(m, dev) if hashed else None) for _ in range(P)]
    for chunks in (1, 2, 4, 8):
        cs, stages = build_column_stages(rowptr, col, [m] * P, 0, chunks, positions if hashed else None)
        pads = []
        for p in range(P):
            xp = torch.empty(chunks * cs, F, device=dev)
            if hashed:
                xp[positions[p]] = shards[p]
            else:
                xp[:m] = shards[p]
            pads.append(xp)
        bufs = [torch.cat([pads[p][c * cs:(c + 1) * cs] for p in range(P)]) for c in range(chunks)]
        vals = [value[st['src']] for st in stages]
        out = torch.empty(m, F, device=dev)

        def run():
            for i, st in enumerate(stages):
                nat.spmm_partial(st['rowptr'], st['col'], vals[i], pads[0] if i == 0 else bufs[i - 1], 'sum', out, None,
                                 None, E, i > 0, None)
        ms = gpu_ms(run, iters=5)
        per = []
        for i, st in enumerate(stages):
            per.append(round(gpu_ms(lambda: nat.spmm_partial(st['rowptr'], st['col'], vals[i], pads[0] if i == 0 else bufs[i - 1],
                                                             'sum', out, None, None, E, i > 0, None), iters=3, warm=1), 3))
        run()
        err = float(((out.double() - ref.double()).abs() / l1.double().clamp(min=1e-30)).max())
        print(json.dumps(dict(mode='staged', hashed=hashed, chunks=chunks, ms=round(ms, 3), per_stage_ms=per,
                              entries=[int(st['src'].numel()) for st in stages], max_err_over_l1=err)), flush=True)
        del pads, bufs, vals, stages
        torch.cuda.empty_cache()

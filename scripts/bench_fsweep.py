"""SpMM-sum time against the feature width, incl. widths that are not a multiple of the 16-byte packet
(class-count sized outputs: 7, 40, 47, 172 ...).  One JSON object per line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_sparse_amd import _native as nat  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')


def gpu_ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        t.append(s.elapsed_time(e))
    t.sort()
    return t[len(t) // 2]


scale = int(os.environ.get('SCALE', 21))
rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
n, E = 1 << scale, c.numel()
v = synth.values(E, device=dev)
widths = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 7, 8, 16, 24, 32, 40, 47, 48, 64, 100, 128, 172]
for dtype in (torch.float32, torch.bfloat16):
    for F in widths:
        x = synth.features(n, F, dtype=dtype, device=dev)
        vv = v.to(dtype)
        ms = gpu_ms(lambda: nat.spmm(rp, c, vv, x, 'sum'))
        es = x.element_size()
        balg = E * (8 + es + F * es) + (n + 1) * 8 + n * F * es
        print(json.dumps(dict(bench='spmm_fsweep', dtype=str(dtype).split('.')[1], F=F, ms=round(ms, 3),
                              gedges=round(E / ms / 1e6, 2), frac_hbm=round(balg / ms / 1e6 / 8000, 3))), flush=True)

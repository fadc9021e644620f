"""Same-box A/B of the relabelled copy of X (TSAMD_SPMM_RELABEL=0 / 1, read on every call) for the
reductions, element types and row sizes where the rule of csrc/spmm.hip:relabel_possible decides.
One JSON object per line: ms with the copy forced off / on and what the default picks."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_sparse_amd import _native as nat, synth  # noqa: E402
from tests.baseline_configs import gpu_ms  # noqa: E402

dev = torch.device('cuda:0')
for scale in (20, 21):
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
    n, E = 1 << scale, c.numel()
    for dtype, widths in ((torch.bfloat16, (64, 128, 256, 512)), (torch.float16, (128, )), (torch.float32, (32, 64, 128, 256))):
        v = synth.values(E, dtype=dtype, device=dev)
        for F in widths:
            x = synth.features(n, F, dtype=dtype, device=dev)
            for reduce, val in (('sum', v), ('max', None), ('max', v)):
                res = dict(scale=scale, dtype=str(dtype).split('.')[1], F=F, row_bytes=F * x.element_size(), reduce=reduce,
                           has_value=val is not None)
                for mode in ('0', '1', None):
                    if mode is None:
                        os.environ.pop('TSAMD_SPMM_RELABEL', None)
                    else:
                        os.environ['TSAMD_SPMM_RELABEL'] = mode
                    res['ms_' + {'0': 'off', '1': 'on', None: 'default'}[mode]] = round(
                        gpu_ms(lambda: nat.spmm(rp, c, val, x, reduce), iters=7, warm=2), 4)
                res['on_over_off'] = round(res['ms_on'] / res['ms_off'], 3)
                print(json.dumps(res), flush=True)
            del x

"""Forward SpMM variants the round-1 verdict flagged (narrow-type min/max, F <= 16): a few launches
each, for rocprofv3 --kernel-trace --stats."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import synth, _native as nat
from tests.baseline_configs import gpu_ms, b_alg
dev = torch.device('cuda:0')
which = sys.argv[1:] or ['c3', 'ns_bf16', 'f16']
def run(tag, scale, K, dtype, red, has_value=True):
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
    v = synth.values(E, dtype=dtype, device=dev) if has_value else None
    x = synth.features(n, K, dtype=dtype, device=dev)
    ms = gpu_ms(lambda: nat.spmm(rp, c, v, x, red), iters=10)
    prof = []
    nat.spmm(rp, c, v, x, red, profile=prof)
    ba = b_alg(E, n, K, x.element_size(), has_value, red in ('min', 'max'))
    print(json.dumps(dict(tag=tag, scale=scale, F=K, dtype=str(dtype).split('.')[1], reduce=red, has_value=has_value, ms=round(ms, 4),
                          frac=round(ba / ms / 1e6 / 8000, 4), pre_ms=round(prof[0], 4), merge_ms=round(prof[1], 4), fixup_ms=round(prof[2], 4))), flush=True)
if 'c3' in which:
    run('c3', 20, 128, torch.bfloat16, 'max', False)
    run('c3', 20, 128, torch.bfloat16, 'max', True)
    run('c3sum', 20, 128, torch.bfloat16, 'sum', True)
if 'ns_bf16' in which:
    run('ns', 21, 128, torch.bfloat16, 'max')
    run('ns', 21, 128, torch.float16, 'max')
    run('ns', 21, 128, torch.float32, 'max')
if 'f16' in which:
    for K in (4, 8, 16, 32):
        run('lowF', 21, K, torch.float32, 'sum')

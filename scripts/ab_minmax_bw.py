"""Same-box A/B of the min/max backward (config 3: 2^20 R-MAT, bf16 F=128): packed narrow atomics vs
fp32 shadow, value-less and with values, plus fp32."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import synth, _native as nat
dev = torch.device('cuda:0')

def gpu_time(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts_ = []
    for _ in range(iters):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts_.append(s.elapsed_time(e))
    ts_.sort()
    return ts_[len(ts_) // 2]

scale, K = 20, 128
rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
for dtype in (torch.bfloat16, torch.float16, torch.float32):
    for has_value in (False, True):
        v = synth.values(E, dtype=dtype, device=dev) if has_value else None
        x = synth.features(n, K, dtype=dtype, device=dev)
        out, arg = nat.spmm(rp, c, v, x, 'max')
        g = synth.features(n, K, seed=3, dtype=dtype, device=dev)
        s = x.element_size()
        bytes_ = n * K * (8 + 2 * s) + 2 * n * K * s
        res = dict(bench='c3_minmax_bw', dtype=str(dtype).split('.')[1], has_value=has_value, alg_bytes=bytes_)
        for mode in ('packed', 'shadow'):
            if dtype == torch.float32 and mode == 'shadow':
                continue
            os.environ['TSAMD_MINMAX_BW_SHADOW'] = '1' if mode == 'shadow' else '0'
            t = gpu_time(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=has_value, want_mat=True))
            res[mode + '_ms'] = round(t, 3)
            res[mode + '_frac'] = round(bytes_ / t / 1e6 / 8000, 3)
            if has_value:
                t = gpu_time(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=False, want_mat=True))
                res[mode + '_matonly_ms'] = round(t, 3)
        # accuracy of the two modes against fp64 scatter
        if dtype != torch.float32:
            invalid = arg == E
            a = arg.masked_fill(invalid, 0)
            ind = c[a]
            w = (v[a].double() if has_value else 1.0)
            ref = torch.zeros(n, K, dtype=torch.float64, device=dev).scatter_add_(0, ind, (w * g.double()).masked_fill(invalid, 0))
            l1 = torch.zeros(n, K, dtype=torch.float64, device=dev).scatter_add_(0, ind, (w * g.double()).abs().masked_fill(invalid, 0))
            for mode in ('packed', 'shadow'):
                os.environ['TSAMD_MINMAX_BW_SHADOW'] = '1' if mode == 'shadow' else '0'
                gv, gm = nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=has_value, want_mat=True)
                err = (gm.double() - ref).abs()
                res[mode + '_max_err_over_l1'] = float((err / l1.clamp(min=1e-30)).max())
                res[mode + '_max_abs_err'] = float(err.max())
            del ref, l1
        print(json.dumps(res), flush=True)
        del x, g, out, arg

"""Same-box A/B of build/ab/*.so variants on the config-3 forward (bf16 max, F = 128, 2^20 R-MAT) and the fp32 max
at north-star size; TSAMD_SPMM_ITEMS as a second axis.  [pre, merge, fix-up] ms, median of 11."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
out = {}
for name, scale, dtype, F in (('c3_bf16_128', 20, torch.bfloat16, 128), ('f16_128', 20, torch.float16, 128), ('ns_f32_128', 21, torch.float32, 128)):
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
    x = synth.features(n, F, dtype=dtype, device=dev)
    for tag, v in (('', None), ('+v', synth.values(c.numel(), device=dev).to(dtype))):
        for _ in range(3): nat.spmm(rp, c, v, x, 'max')
        rows = []
        for _ in range(11):
            prof = []
            nat.spmm(rp, c, v, x, 'max', profile=prof)
            rows.append(prof)
        med = [sorted(r[i] for r in rows)[5] for i in range(3)]
        out[name + tag] = [round(m, 4) for m in med] + [round(sum(med), 4)]
print(json.dumps(out))
''' % ROOT
for items in ((None, '512') if os.environ.get('AB_ITEMS') else (None,)):
    for name in sys.argv[1:]:
        env = dict(os.environ, TSAMD_LIB=os.path.join(ROOT, 'build', 'ab', name + '.so'))
        if items:
            env['TSAMD_SPMM_ITEMS'] = items
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.split('\n') if l.startswith('{')]
        print(name, 'items', items, line[-1] if line else out.stderr[-300:], flush=True)

"""Same-box A/B of build/ab/*.so variants on the three SpMM shapes whose fix-up launch matters: north star (sum fp32
F = 128, 2^21), config 2 (sum fp32 F = 64, 2^20) and config 3 (max bf16 F = 128, 2^20).  [pre, merge, fix-up, sum] ms,
median of 15, through the C-ABI (TSAMD_LIB).  Usage: python scripts/ab_fixup.py base fix2"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
out = {}
import os
only = os.environ.get('AB_SHAPES')
for name, scale, dtype, F, red in (('ns_sum_f32_128', 21, torch.float32, 128, 'sum'), ('c2_sum_f32_64', 20, torch.float32, 64, 'sum'),
                                   ('c3_max_bf16_128', 20, torch.bfloat16, 128, 'max'), ('c5_sum_f32_256', 20, torch.float32, 256, 'sum'),
                                   ('c3v_max_bf16_128', 20, torch.bfloat16, 128, 'max'), ('f16_min_f16_64', 20, torch.float16, 64, 'min'),
                                   ('nsmax_max_f32_128', 21, torch.float32, 128, 'max')):
    if only and name.split('_')[0] not in only.split(','): continue
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale
    x = synth.features(n, F, dtype=dtype, device=dev)
    v = synth.values(c.numel(), dtype=dtype, device=dev) if name.startswith('c3v') else None
    for _ in range(3): nat.spmm(rp, c, v, x, red)
    rows = []
    for _ in range(15):
        prof = []
        nat.spmm(rp, c, v, x, red, profile=prof)
        rows.append(prof)
    med = [sorted(r[i] for r in rows)[7] for i in range(3)]
    out[name] = [round(m, 4) for m in med] + [round(sum(med), 4)]
print(json.dumps(out))
''' % ROOT
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, TSAMD_LIB=os.path.join(ROOT, 'build', 'ab', name + '.so'))
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.split('\n') if l.startswith('{')]
        print(name, line[-1] if line else out.stderr[-300:], flush=True)

"""Same-box A/B of spmm_permute_rows_kernel variants (build/ab/*.so via TSAMD_LIB, one subprocess each):
north-star call through the C-ABI, [pre (probe + copy + partition), merge, fix-up] ms, median of 15."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
rp, c = synth.rmat_csr(21, 20, seed=0, device=dev); n = 1 << 21
v = synth.values(c.numel(), device=dev); x = synth.features(n, 128, device=dev)
for _ in range(3): nat.spmm(rp, c, v, x, 'sum')
rows = []
for _ in range(15):
    prof = []
    nat.spmm(rp, c, v, x, 'sum', profile=prof)
    rows.append(prof)
med = [sorted(r[i] for r in rows)[7] for i in range(3)]
print(json.dumps(dict(pre=round(med[0], 4), merge=round(med[1], 4), fixup=round(med[2], 4), total=round(sum(med), 4))))
''' % ROOT
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, TSAMD_LIB=os.path.join(ROOT, 'build', 'ab', name + '.so'))
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.split('\n') if l.startswith('{')]
        print(name, line[-1] if line else out.stderr[-300:], flush=True)

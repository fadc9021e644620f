"""A/B of build/ab/sort_*.so variants of the one-sweep sort (ctypes path): tsamd_sort_coo on 7.5 M uniform draws
over 500k x 500k, median HIP-event time.  Build the variants with
    python scripts/exp_sort_variants.py --build name:DEF1,DEF2 ...     (here, no GPU needed)
run with  python scripts/exp_sort_variants.py                          (GPU box)"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, 'build', 'ab')
CSRC = os.path.join(ROOT, 'pytorch_sparse_amd', 'csrc')
if '--build' in sys.argv:
    os.makedirs(AB, exist_ok=True)
    procs = []
    for a in sys.argv[sys.argv.index('--build') + 1:]:
        name, _, d = a.partition(':')
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
               '-I' + CSRC] + ['-D' + x for x in d.split(',') if x] + [os.path.join(CSRC, f) for f in ('api.hip', 'sort.hip', 'coalesce.hip')] + \
              ['-o', os.path.join(AB, 'sort_%s.so' % name)]
        procs.append(subprocess.Popen(cmd))
    sys.exit(max(p.wait() for p in procs))

import torch  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402
dev = torch.device('cuda:0')
m = n = int(os.environ.get('MN', 500000))
E = int(os.environ.get('E', 7500000))
row, col = synth.uniform_edges(m, n, E, seed=0, device=dev)
i64 = ctypes.c_int64
for so in sorted(glob.glob(os.path.join(AB, 'sort_*.so'))):
    L = ctypes.CDLL(so, mode=os.RTLD_LAZY)  # (tsamd_gather_rows of the one-launch path lives in another TU)
    L.tsamd_sort_coo_workspace_bytes.restype = ctypes.c_size_t
    ws = torch.empty(L.tsamd_sort_coo_workspace_bytes(i64(E)), dtype=torch.uint8, device=dev)
    ro, co, po = torch.empty_like(row), torch.empty_like(col), torch.empty_like(row)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        st = L.tsamd_sort_coo(ctypes.c_void_p(row.data_ptr()), ctypes.c_void_p(col.data_ptr()), i64(E), i64(m), i64(n),
                              ctypes.c_void_p(ro.data_ptr()), ctypes.c_void_p(co.data_ptr()), ctypes.c_void_p(po.data_ptr()),
                              ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), stream)
        assert st == 0, st
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    ok = bool((ro[1:] * n + co[1:] >= ro[:-1] * n + co[:-1]).all())
    print(json.dumps(dict(variant=os.path.basename(so), ms=round(ts[len(ts) // 2], 4), min_ms=round(ts[0], 4), sorted=ok)), flush=True)

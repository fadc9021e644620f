import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
rp, c = synth.rmat_csr(21, 20, seed=0, device=dev); n = 1 << 21; E = c.numel()
for dtype, K in ((torch.bfloat16, 128), (torch.float16, 128), (torch.bfloat16, 256), (torch.bfloat16, 64)):
    v = synth.values(E, dtype=dtype, device=dev); x = synth.features(n, K, dtype=dtype, device=dev)
    for red in ('sum', 'max'):
        for _ in range(3): nat.spmm(rp, c, v, x, red)
        ts = []
        for _ in range(7):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); nat.spmm(rp, c, v, x, red); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
        ts.sort(); print(os.environ.get('TSAMD_LIB', '')[-6:], str(dtype)[6:], K, red, '%.3f ms' % ts[3], flush=True)

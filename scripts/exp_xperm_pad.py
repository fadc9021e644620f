"""Does the POSITION of the relabelled copy of X inside the workspace matter?  TSAMD_SPMM_XPERM_PAD shifts it; north-star
shape on two R-MAT seeds; merge-kernel ms (median of 9) per shift.  Through the C-ABI (TSAMD_LIB or the shipped library)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_sparse_amd import _native as nat, synth
dev = torch.device('cuda:0')
pads = [0, 256, 512, 1024, 2048, 4096, 8192, 65536, 1 << 20, 1325568, 2 << 20, (2 << 20) + 256, 5 << 20, 0]
for seed in (0, 1):
    rp, c = synth.rmat_csr(21, 20, seed=seed, device=dev)
    x = synth.features(1 << 21, 128, dtype=torch.float32, device=dev)
    row = {}
    for pad in pads:
        os.environ['TSAMD_SPMM_XPERM_PAD'] = str(pad)
        for _ in range(2): nat.spmm(rp, c, None, x, 'sum')
        t = []
        for _ in range(9):
            prof = []
            nat.spmm(rp, c, None, x, 'sum', profile=prof)
            t.append(prof[1])
        row.setdefault(str(pad), []).append(round(sorted(t)[4], 4))
    print(json.dumps({'seed': seed, 'merge_ms_by_pad': row}), flush=True)

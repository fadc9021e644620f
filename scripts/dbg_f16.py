import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pytorch_sparse_amd import _native as nat, synth
from tests.util import oracle_spmm
from tests.test_spmm_gpu import make_inputs
dev = torch.device('cuda:0')
rp, c = synth.rmat_csr(10, 16, seed=0)
for K, hv, batch in ((128, True, ()), (16, False, ()), (3, True, (2,)), (64, True, ())):
    v, x = make_inputs(rp, c, 1024, K, torch.float16, hv, batch)
    out, arg = nat.spmm(rp.to(dev), c.to(dev), None if v is None else v.to(dev), x.to(dev), 'max')
    eo, ea = oracle_spmm(rp, c, v, x, 'max')
    o = out.cpu()
    bad = (o.view(torch.int16) != eo.view(torch.int16)).nonzero()
    print(K, hv, batch, 'nbad', len(bad), 'arg equal', torch.equal(arg.cpu(), ea))
    for idx in bad[:5]:
        idx = tuple(idx.tolist())
        a = int(arg.cpu()[idx]); 
        print('  at', idx, 'gpu', float(o[idx]), hex(o.view(torch.int16)[idx].item() & 0xffff), 'oracle', float(eo[idx]), hex(eo.view(torch.int16)[idx].item() & 0xffff), 'arg', a, int(ea[idx]),
              'v', None if v is None else float(v[a]), 'x', float(x[(*idx[:-2], int(c[a]), idx[-1])]))

"""Stress / C4 SpSpMM time of the library that is loaded (LD_PRELOAD picks the variant) -> one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_sparse_amd  # noqa: E402,F401
from tests import baseline_configs as bc  # noqa: E402

torch.set_num_threads(32)
dev = torch.device('cuda:0')
out = dict(variant=os.environ.get('VARIANT', 'shipped'))
for kind in sys.argv[1:] or ['stress']:
    A, At = bc.spspmm_inputs(dev, kind)
    ms = bc.gpu_ms(lambda: A @ At, iters=5, warm=2)
    out[kind + '_ms'] = round(ms, 3)
    if kind == 'stress' and os.environ.get('CHECK'):
        C = A @ At
        out['parity_ok'] = bc.spspmm_properties(A, At, C)['ok']
print(json.dumps(out), flush=True)

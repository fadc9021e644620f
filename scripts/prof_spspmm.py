"""C4 / R-MAT stress SpSpMM timing (whole op) -- run under rocprofv3 --kernel-trace --stats for the breakdown."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import baseline_configs as bc
dev = torch.device('cuda:0')
import pytorch_sparse_amd  # noqa
for kind in sys.argv[1:] or ['c4']:
    r = bc.run_spspmm(dev, kind, cpu='cpu' in os.environ.get('SPSPMM_CHECK', ''), iters=5)
    print(json.dumps(r), flush=True)

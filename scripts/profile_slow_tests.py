"""cProfile of the slowest GPU tests' bodies on the GPU box (round 5: where does the host time go?)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_sparse_amd  # noqa: E402,F401
from tests import test_sort_gpu, test_spmm_gpu  # noqa: E402

dev = torch.device('cuda:0')
torch.set_num_threads(int(os.environ.get('NT', '32')))
for name, fn in (('winner_lists[f32]', lambda: test_spmm_gpu.test_minmax_bw_winner_lists_hub_columns(dev, torch.float32)),
                 ('onesweep_power_law', lambda: test_sort_gpu.test_onesweep_sort_power_law_and_hot_digits(dev, torch.ops.tsamd)),
                 ('ind2ptr_ptr2ind', lambda: test_spmm_gpu.test_ind2ptr_ptr2ind(dev))):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    fn()
    pr.disable()
    print('=== %s: %.1f s' % (name, time.perf_counter() - t0), flush=True)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)

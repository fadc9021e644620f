import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _bounded_host_threads():
    """ATen's host ops with one thread per core of a 256-core box are 10-80x SLOWER on the test-sized tensors of this
    suite than with a few dozen threads (measured on the GPU box, scripts/profile_slow_tests.py: three tests at
    25-31 s each took 0.2-1.8 s with 32 threads); the full-size config rows raise the count themselves around their
    timed CPU-baseline legs (tests/baseline_configs.py), so every test starts from a bounded pool."""
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    yield


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once so that the
    suite does not depend on an earlier build() call.  No-op when everything is in place."""
    lib = os.path.join(ROOT, 'pytorch_sparse_amd', 'lib')
    need = [os.path.join(lib, 'libtsamd.so'), os.path.join(lib, '_tsamd_ops.so'),
            os.path.join(ROOT, 'oracle', 'libts_oracle.so')]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__
    __graft_entry__.build()

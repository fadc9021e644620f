"""672 randomised small cases of the whole Python surface (narrow / index_select / masked_select / diag /
cat / permute / element-wise / reductions / transpose / coalesce / to_symmetric / __getitem__ / SAINT /
take-all sample_adj / matmul forward and backward / the functional spmm, coalesce, transpose, spspmm) against the outputs of the REFERENCE package on the same inputs
(tests/golden/py6_random_cases.npz, written by make_golden.py part 6 from tests/golden/cases6.py)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
sys.path.insert(0, GOLDEN)
import cases6  # noqa: E402

_BLOB = None


def _blob():
    global _BLOB
    if _BLOB is None:
        z = np.load(os.path.join(GOLDEN, 'py6_random_cases.npz'))
        _BLOB = {k: z[k] for k in z.files}
    return _BLOB


def _case(i):
    b = _blob()
    pre = 'c%d_' % i
    c = {k[len(pre):]: v for k, v in b.items() if k.startswith(pre)}
    outs = [b['o%d_%d' % (i, j)] for j in range(int(b['n%d' % i]))]
    return c, outs


N_CASES = 672


@pytest.mark.parametrize('chunk', range(0, N_CASES, 28))
def test_random_cases_match_the_reference(chunk):
    import pytorch_sparse_amd as ts
    for i in range(chunk, min(chunk + 28, N_CASES)):
        if ('n%d' % i) not in _blob():
            continue  # the reference raised for this draw
        c, want = _case(i)
        c = {k: (v.item() if v.ndim == 0 and v.dtype.kind in 'iuUb' else v) for k, v in c.items()}
        got = cases6.run_case(ts, c, 'cuda')
        tag = 'case %d (%s)' % (i, c['op'])
        assert len(got) == len(want), tag
        for g, w in zip(got, want):
            assert g.shape == w.shape, '%s: shape %s vs %s' % (tag, g.shape, w.shape)
            if w.dtype.kind == 'f':
                np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-6, err_msg=tag)
            else:
                np.testing.assert_array_equal(g, w, err_msg=tag)

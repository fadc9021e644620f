"""BASELINE.json configs C2 / C3 / C4 at their STATED size, whole-output parity:
the same functions bench.py puts into the `secondary` array of its JSON line."""
import pytest
import torch

from tests import baseline_configs as bc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import pytorch_sparse_amd  # noqa: F401  registers torch.ops.torch_sparse.*
    return torch.ops.torch_sparse


def test_c2_sum_f32_full_output(dev, ops):
    """2^20 R-MAT, F = 64 fp32: all 67 M outputs against the compiled reference CPU kernel."""
    r = bc.run_c2(dev, iters=3)
    p = r['parity']
    assert p['elements'] == (1 << 20) * 64
    assert p['max_err_over_l1'] <= 1e-5 and p['ours_vs_fp64_over_l1'] <= 1e-5, p
    # not less accurate than the reference's own sequential fp32 sum
    assert p['ours_vs_fp64_over_l1'] <= max(p['ref_vs_fp64_over_l1'], 2e-7) * 1.5, p


@pytest.mark.parametrize('has_value', [False, True])
def test_c3_max_bf16_fwd_bwd_full_size(dev, ops, has_value):
    """2^20 R-MAT, F = 128 bf16, max + backward: out and arg_out bit-exact over all 134 M elements
    against the reference CPU kernel; grad_mat / grad_value within the rounding bound of their
    arithmetic against the fp64 formulas of csrc/spmm.cpp:204-242."""
    r = bc.run_c3(dev, has_value, iters=2)
    p = r['parity']
    assert p['elements'] == (1 << 20) * 128
    assert p['arg_out_mismatches'] == 0 and p['out_bit_mismatches'] == 0, p
    assert p['grad_mat_max_err_over_bound'] <= 1.0 and p['grad_mat_autograd_equal_bound'] <= 1.0, p
    if has_value:
        assert p['grad_value_max_err_over_bound'] <= 1.0 and p['grad_value_autograd_max_err_over_bound'] <= 1.0, p
    assert p['ok']


def test_c4_spspmm_full_size(dev, ops):
    """A * A^T, 500k x 500k, 7.5 M draws: (row, col) bit-exact against torch.sparse.mm on the host
    (what torch_sparse/matmul.py:104 calls), values within 1e-5 of the L1 mass."""
    r = bc.run_spspmm(dev, 'c4', iters=2)
    p = r['parity']
    assert p['index_bit_exact'], p
    assert p['value_max_err_over_l1'] <= 1e-5, p


def test_spspmm_rmat_stress_row(dev, ops):
    """SURVEY 8d stress row: A * A^T of an R-MAT scale-19 graph (2.3 G products, hub rows of 10^7 products go
    through the binned dense accumulation).  torch.sparse.mm on the host would take minutes, so: structure,
    fp64 checksums of every row, and 49 rows (incl. the longest) compared exactly with torch.sparse.mm."""
    r = bc.run_spspmm(dev, 'stress', cpu=False, iters=1)
    p = r['parity']
    assert p['rows_sorted_unique'] and p['sampled_rows_index_bit_exact'], p
    assert p['row_sum_max_err_over_l1'] <= 1e-5 and p['sampled_rows_value_max_err_over_l1'] <= 1e-5, p
    assert p['ok']

"""BASELINE.json configs C1 / C2 / C3 / C4 at their STATED size plus the construct / coalesce / transpose and
backward rows, whole-output parity: the same functions bench.py puts into the `secondary` array of its JSON line."""
import pytest
import torch

from tests import baseline_configs as bc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import pytorch_sparse_amd  # noqa: F401  registers torch.ops.torch_sparse.*
    return torch.ops.torch_sparse


def test_c1_legacy_spmm_exact_config(dev, ops):
    """configs[0]: torch_sparse.spmm(index, value, 1000, 1000, x) on 5 000 unsorted draws, F = 16 fp32, against
    the output the reference's own Python produced for these inputs (tests/golden/py7_c1_spmm.npz)."""
    r = bc.run_c1(dev, iters=5)
    p = r['parity']
    assert p['elements'] == 1000 * 16
    assert p['max_err_over_l1'] <= 1e-5 and p['vs_fp64_over_l1'] <= 1e-5, p
    assert p['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0, p
    assert r['cpu_baseline']['port_matches_fixture']
    assert p['ok']


def test_construct_coalesce_transpose_7m5(dev, ops):
    """a9-a12 on the 7.5 M-entry configs[3] input: sorted (row, col), rowptr, coalesced / transposed index and the
    t() permutation bit-exact against the numpy restatement of the reference Python (independent of the product's sort)."""
    r = bc.run_construct(dev, iters=2)
    p = r['parity']
    for k in ('construct_row_col_bit_exact', 'construct_value_bit_exact', 'construct_rowptr_bit_exact',
              'coalesce_index_bit_exact', 'transpose_index_bit_exact', 't_row_col_bit_exact', 't_value_bit_exact',
              'csr2csc_bit_exact'):
        assert p[k], (k, p)
    assert p['coalesce_value_max_abs_err'] <= 1e-6 and p['transpose_value_max_abs_err'] <= 1e-6, p
    assert p['ok']


def test_c2_value_grad_and_sum_fw_bw(dev, ops):
    """a3 / a4 at config-2 size: grad_value and grad_mat of adj.matmul(x).backward(g) against the compiled
    reference's autograd op, whole outputs, 1e-5 of the L1 mass of each sum."""
    r = bc.run_c2_backward(dev, iters=2)
    p = r['parity']
    assert p['grad_value_vs_fp64_over_l1'] <= 1e-5 and p['grad_value_vs_ref_over_l1'] <= 1e-5, p
    assert p['grad_mat']['ok'] and p['grad_mat']['elements'] == (1 << 20) * 64, p
    assert p['forward_max_err_over_l1'] <= 1e-5, p
    assert p['ok']


def test_c2_sum_f32_full_output(dev, ops):
    """2^20 R-MAT, F = 64 fp32: all 67 M outputs against the compiled reference CPU kernel."""
    r = bc.run_c2(dev, iters=3)
    p = r['parity']
    assert p['elements'] == (1 << 20) * 64
    assert p['max_err_over_l1'] <= 1e-5 and p['ours_vs_fp64_over_l1'] <= 1e-5, p
    # element-wise 1e-5 relative wherever the sum is well conditioned (|ref| >= 0.1 * L1 mass)
    assert p['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0, p
    # not less accurate than the reference's own sequential fp32 sum
    assert p['ours_vs_fp64_over_l1'] <= max(p['ref_vs_fp64_over_l1'], 2e-7) * 1.5, p


def test_c5_per_gpu_share_full_output_and_8_logical_ranks(dev, ops):
    """configs[4]'s per-GPU share (2^21 rows, ~32 nnz/row, F = 256 fp32: 1 KB rows, the 1024-item partition branch):
    all 537 M outputs against the compiled reference CPU kernel, and the row-sharded product with P = 8 logical
    ranks on this one device against the unsharded one (max + arg bit for bit, sum to 1e-5 of the L1 mass)."""
    r = bc.run_c5_share(dev, iters=3, fp64_leg=False)  # (the fp64 statistics of this shape: bench.py's c5_share row)
    p = r['parity']
    assert p['elements'] == (1 << 21) * 256
    assert p['max_err_over_l1'] <= 1e-5, p
    assert p['n_rel_gt_1e_5_where_ref_ge_1e_1_l1'] == 0, p
    rs = r['row_sharded']
    assert rs['max_and_arg_bit_identical_to_unsharded'] and rs['sum_max_err_over_l1_vs_unsharded'] <= 1e-5, rs
    assert p['ok']


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

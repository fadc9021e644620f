"""The operator library registers the reference's op names with the reference's schemas
(csrc/spmm.cpp:344-348, csrc/convert.cpp:46-48, csrc/version.cpp:40-41).
Loading the library needs no GPU."""
import torch

import pytorch_sparse_amd  # noqa: F401

EXPECTED = {
    'spmm_sum': 'torch_sparse::spmm_sum(Tensor? _0, Tensor _1, Tensor _2, Tensor? _3, Tensor? _4, Tensor? _5, Tensor _6) -> Tensor _0',
    'spmm_mean': 'torch_sparse::spmm_mean(Tensor? _0, Tensor _1, Tensor _2, Tensor? _3, Tensor? _4, Tensor? _5, Tensor? _6, Tensor _7) -> Tensor _0',
    'spmm_min': 'torch_sparse::spmm_min(Tensor _0, Tensor _1, Tensor? _2, Tensor _3) -> (Tensor _0, Tensor _1)',
    'spmm_max': 'torch_sparse::spmm_max(Tensor _0, Tensor _1, Tensor? _2, Tensor _3) -> (Tensor _0, Tensor _1)',
    'ind2ptr': 'torch_sparse::ind2ptr(Tensor _0, int _1) -> Tensor _0',
    'ptr2ind': 'torch_sparse::ptr2ind(Tensor _0, int _1) -> Tensor _0',
    'cuda_version': 'torch_sparse::cuda_version() -> int _0',
    # csrc/diag.cpp:22-36 (widening)
    'non_diag_mask': 'torch_sparse::non_diag_mask(Tensor _0, Tensor _1, int _2, int _3, int _4) -> Tensor _0',
    # csrc/rw.cpp, sample.cpp, relabel.cpp, saint.cpp (widening)
    'random_walk': 'torch_sparse::random_walk(Tensor _0, Tensor _1, Tensor _2, int _3) -> Tensor _0',
    'sample_adj': 'torch_sparse::sample_adj(Tensor _0, Tensor _1, Tensor _2, int _3, bool _4) -> (Tensor _0, Tensor _1, Tensor _2, Tensor _3)',
    'relabel': 'torch_sparse::relabel(Tensor _0, Tensor _1) -> (Tensor _0, Tensor _1)',
    'relabel_one_hop': 'torch_sparse::relabel_one_hop(Tensor _0, Tensor _1, Tensor? _2, Tensor _3, bool _4) -> (Tensor _0, Tensor _1, Tensor? _2, Tensor _3)',
    'saint_subgraph': 'torch_sparse::saint_subgraph(Tensor _0, Tensor _1, Tensor _2, Tensor _3) -> (Tensor _0, Tensor _1, Tensor _2)',
    'neighbor_sample': 'torch_sparse::neighbor_sample(Tensor _0, Tensor _1, Tensor _2, int[] _3, bool _4, bool _5) -> (Tensor _0, Tensor _1, Tensor _2, Tensor _3)',
    # csrc/neighbor_sample.cpp:29-63 (heterogeneous / temporal multi-hop sampling)
    'hetero_neighbor_sample': 'torch_sparse::hetero_neighbor_sample(str[] _0, (str, str, str)[] _1, Dict(str, Tensor) _2, Dict(str, Tensor) _3, Dict(str, Tensor) _4, Dict(str, int[]) _5, int _6, bool _7, bool _8) -> (Dict(str, Tensor) _0, Dict(str, Tensor) _1, Dict(str, Tensor) _2, Dict(str, Tensor) _3)',
    'hetero_temporal_neighbor_sample': 'torch_sparse::hetero_temporal_neighbor_sample(str[] _0, (str, str, str)[] _1, Dict(str, Tensor) _2, Dict(str, Tensor) _3, Dict(str, Tensor) _4, Dict(str, int[]) _5, Dict(str, Tensor) _6, int _7, bool _8, bool _9) -> (Dict(str, Tensor) _0, Dict(str, Tensor) _1, Dict(str, Tensor) _2, Dict(str, Tensor) _3)',
}


def test_reference_op_schemas():
    for name, schema in EXPECTED.items():
        op = getattr(torch.ops.torch_sparse, name)
        assert str(op.default._schema) == schema, (name, str(op.default._schema))
    assert torch.ops.torch_sparse.cuda_version() >= 60000000


def test_schemas_equal_the_compiled_reference():
    """Every op registered here under torch_sparse:: has exactly the schema the reference's own op
    file registers (its csrc/*.cpp compiled unmodified into oracle/_ref under the ts_ref:: namespace)."""
    import pytest
    from oracle import ref
    if not ref.available():
        pytest.skip('oracle/_ref not built')
    ref.ops()
    for name in EXPECTED:
        if name == 'cuda_version':  # csrc/version.cpp is not part of oracle/_ref
            continue
        want = str(getattr(torch.ops.ts_ref, name).default._schema).replace('ts_ref::', 'torch_sparse::')
        assert str(getattr(torch.ops.torch_sparse, name).default._schema) == want, name


def test_ops_refuse_cpu_tensors():
    import pytest
    rowptr, col, x = torch.tensor([0, 1]), torch.tensor([0]), torch.ones(1, 4)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        torch.ops.torch_sparse.spmm_sum(None, rowptr, col, None, None, None, x)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        torch.ops.torch_sparse.ind2ptr(col, 3)
    # unsorted COO on the CPU is accepted but stays locked until it is moved to the GPU
    A = pytorch_sparse_amd.SparseTensor(row=torch.tensor([1, 0]), col=torch.tensor([0, 0]))
    assert A.nnz() == 2 and A.sparse_sizes() == (2, 1)
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        A.coo()
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        A.storage.rowptr()
    # sorted data on the CPU can be held (and inspected), it just cannot be computed on
    B = pytorch_sparse_amd.SparseTensor(row=torch.tensor([0, 1]), col=torch.tensor([0, 0]), is_sorted=True)
    assert B.storage.col().tolist() == [0, 0]
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        B.storage.rowptr()


def test_api_surface():
    for name in ('SparseStorage', 'SparseTensor', 'matmul', 'spmm', 'spspmm', 'coalesce', 'transpose', 't'):
        assert hasattr(pytorch_sparse_amd, name)
    for m in ('matmul', 'spmm', 'spspmm', 't', 'coalesce', 'csr', 'coo', 'csc', 'to_dense'):
        assert hasattr(pytorch_sparse_amd.SparseTensor, m)
    # the widened surface: every name the reference package exports for it
    for name in ('narrow', '__narrow_diag__', 'select', 'index_select', 'index_select_nnz', 'masked_select',
                 'masked_select_nnz', 'permute', 'remove_diag', 'set_diag', 'fill_diag', 'get_diag', 'sample', 'sample_adj', 'random_walk', 'saint_subgraph', 'add',
                 'add_', 'add_nnz', 'add_nnz_', 'mul', 'mul_', 'mul_nnz', 'mul_nnz_', 'sum', 'mean', 'min',
                 'max', 'cat', 'to_torch_sparse', 'from_torch_sparse', 'to_scipy', 'from_scipy', 'eye', 'spadd'):
        assert hasattr(pytorch_sparse_amd, name), name
    for m in ('narrow', 'select', 'index_select', 'index_select_nnz', 'masked_select', 'masked_select_nnz',
              'permute', 'remove_diag', 'set_diag', 'fill_diag', 'get_diag', 'add', 'add_', 'mul', 'mul_',
              '__getitem__', '__add__', '__mul__', 'sum', 'mean', 'min', 'max', 'sample', 'sample_adj', 'random_walk',
              'saint_subgraph'):
        assert hasattr(pytorch_sparse_amd.SparseTensor, m), m


def test_every_name_the_reference_exports_is_there():
    """torch_sparse/__init__.py:68-113 (`__all__`), verbatim."""
    ref_all = ['SparseStorage', 'SparseTensor', 't', 'narrow', '__narrow_diag__', 'select', 'index_select',
               'index_select_nnz', 'masked_select', 'masked_select_nnz', 'permute', 'remove_diag', 'set_diag',
               'fill_diag', 'get_diag', 'add', 'add_', 'add_nnz', 'add_nnz_', 'mul', 'mul_', 'mul_nnz', 'mul_nnz_',
               'sum', 'mean', 'min', 'max', 'matmul', 'cat', 'random_walk', 'partition', 'reverse_cuthill_mckee',
               'saint_subgraph', 'to_torch_sparse', 'from_torch_sparse', 'to_scipy', 'from_scipy', 'coalesce',
               'transpose', 'eye', 'spmm', 'spspmm', 'spadd', '__version__']
    missing = [n for n in ref_all if not hasattr(pytorch_sparse_amd, n)]
    assert not missing, missing
    import pytest
    B = pytorch_sparse_amd.SparseTensor(row=torch.tensor([0, 1]), col=torch.tensor([0, 0]), is_sorted=True)
    assert pytorch_sparse_amd.partition(B, 1)[0] is B
    with pytest.raises(RuntimeError, match='METIS'):
        pytorch_sparse_amd.partition(B, 2)


def test_view_ops_need_no_kernel():
    """narrow(0) / cat(0) / cat((0,1)) on a CSR-holding tensor are views and memcpys: they work on host
    tensors too (that is how a loader process cuts the row shards before the upload)."""
    ts = pytorch_sparse_amd
    a = ts.SparseTensor(rowptr=torch.tensor([0, 2, 3, 5]), col=torch.tensor([0, 2, 1, 0, 2]),
                        value=torch.arange(5.), sparse_sizes=(3, 3), is_sorted=True)
    b = a.narrow(0, 1, 2)
    assert b.sparse_sizes() == (2, 3) and b.storage.rowptr().tolist() == [0, 1, 3]
    assert b.storage.col().tolist() == [1, 0, 2] and b.storage.value().tolist() == [2., 3., 4.]
    c = ts.cat([a.narrow(0, 0, 1), b], 0)
    assert c.storage.rowptr().tolist() == [0, 2, 3, 5] and c.storage.col().tolist() == [0, 2, 1, 0, 2]
    d = ts.cat([a, b], (0, 1))
    assert d.sparse_sizes() == (5, 6) and d.storage.col().tolist() == [0, 2, 1, 0, 2, 4, 3, 5]
    assert a[1:].storage.col().tolist() == [1, 0, 2] and a[-1].storage.value().tolist() == [3., 4.]
    import pytest
    with pytest.raises(RuntimeError, match='no CPU implementation'):
        a.narrow(1, 0, 2)

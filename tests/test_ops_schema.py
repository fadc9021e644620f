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
}


def test_reference_op_schemas():
    for name, schema in EXPECTED.items():
        op = getattr(torch.ops.torch_sparse, name)
        assert str(op.default._schema) == schema, (name, str(op.default._schema))
    assert torch.ops.torch_sparse.cuda_version() >= 60000000


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

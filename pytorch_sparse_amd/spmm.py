"""Legacy functional SpMM ``spmm(index, value, m, n, matrix)`` (reference: torch_sparse/spmm.py).

The reference gathers ``matrix[col] * value`` into an ``[nnz, F]`` temporary and scatter-adds it.
Here the COO is ordered once (stable radix sort, duplicates kept -- they add up, as in the
reference) and the CSR SpMM kernel does the rest; autograd w.r.t. ``value`` and ``matrix`` comes
from the op's own backward kernels.
"""
import torch
from torch import Tensor


def spmm(index: Tensor, value: Tensor, m: int, n: int, matrix: Tensor) -> Tensor:
    assert n == matrix.size(-2)
    row, col = index[0], index[1]
    matrix = matrix if matrix.dim() > 1 else matrix.unsqueeze(-1)
    value = value.to(matrix.dtype) if value.dtype != matrix.dtype else value
    nnz = col.numel()
    perm = None
    if nnz > 1 and int(torch.ops.tsamd.coo_order(row, col, n)[0]) > 0:
        row, col, perm = torch.ops.tsamd.sort_coo(row, col, m, n, True)
        value = value[perm]  # differentiable gather
    rowptr = torch.ops.torch_sparse.ind2ptr(row, m)
    need_csc = matrix.requires_grad
    colptr = csr2csc = None
    if need_csc:
        _, _, csr2csc = torch.ops.tsamd.sort_coo(col, row, n, m, False)
        colptr = torch.ops.torch_sparse.ind2ptr(col[csr2csc], n)
    return torch.ops.torch_sparse.spmm_sum(row, rowptr, col, value, colptr, csr2csc, matrix)

"""Legacy functional SpMM ``spmm(index, value, m, n, matrix)`` (reference: torch_sparse/spmm.py).

The reference gathers ``matrix[col] * value`` into an ``[nnz, F]`` temporary and scatter-adds it.
Here the COO is ordered once (stable radix sort decided on the device, duplicates kept -- they add up,
as in the reference) and the CSR SpMM kernel does the rest, without a host sync; autograd w.r.t. ``value`` and ``matrix`` comes
from the op's own backward kernels.
"""
from typing import Optional

import torch
from torch import Tensor


def spmm(index: Tensor, value: Tensor, m: int, n: int, matrix: Tensor) -> Tensor:
    assert n == matrix.size(-2)
    row, col = index[0], index[1]
    matrix = matrix if matrix.dim() > 1 else matrix.unsqueeze(-1)
    value = value.to(matrix.dtype) if value.dtype != matrix.dtype else value
    nnz = col.numel()
    # small inputs, nothing to differentiate: ONE launch on the unsorted COO (tsamd_spmm_coo_small: workgroups own row
    # ranges of `out` in LDS and scan the entries) -- what the reference's three ATen calls do, without the [nnz, F]
    # temporary; the sorted route below costs 8+ launches whatever the size
    wants_grad = torch.is_grad_enabled() and (value.requires_grad or matrix.requires_grad)
    # (its floating-point sums are accumulated through LDS atomics in arrival order: not taken when deterministic
    # algorithms are asked for -- the sorted route below adds a row's entries in a fixed order)
    if (matrix.is_cuda and not wants_grad and m * matrix.size(-1) > 0 and
            not (matrix.is_floating_point() and torch.ops.tsamd.deterministic())):
        out = torch.ops.tsamd.spmm_coo_small(index, value, m, n, matrix)  # (an empty tensor: not taken)
        if out.dim() == 2:
            return out
    if nnz > 1:
        # ordered on the device without asking the host (tsamd::sort_coo_auto: a sorted input only pays the
        # probe and a copy): the whole call enqueues kernels and returns -- no sync
        row, col, perm, _ = torch.ops.tsamd.sort_coo_auto(row.contiguous(), col.contiguous(), m, n)
        value = value.index_select(0, perm)  # differentiable gather (the identity when the input was in order)
    rowptr = torch.ops.torch_sparse.ind2ptr(row, m)
    need_csc = matrix.requires_grad
    colptr: Optional[Tensor] = None
    csr2csc: Optional[Tensor] = None
    if need_csc:
        _, _, csr2csc = torch.ops.tsamd.sort_coo(col, row, n, m, False)
        colptr = torch.ops.torch_sparse.ind2ptr(col[csr2csc], n)
    return torch.ops.torch_sparse.spmm_sum(row, rowptr, col, value, colptr, csr2csc, matrix)

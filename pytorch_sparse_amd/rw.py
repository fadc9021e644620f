"""Uniform random walks (reference: torch_sparse/rw.py, csrc/cpu/rw_cpu.cpp, csrc/cuda/rw_cuda.cu).
``torch.ops.torch_sparse.random_walk`` keeps the reference's schema; ``torch.ops.tsamd.
random_walk_with_rand`` takes the uniform floats as an input, which makes the walk a pure function of
its arguments (that is what the parity tests pin against the reference's CPU kernel)."""
import torch
from torch import Tensor

from .tensor import SparseTensor


def random_walk(src: SparseTensor, start: Tensor, walk_length: int) -> Tensor:
    rowptr, col, _ = src.csr()
    return torch.ops.torch_sparse.random_walk(rowptr, col, start, walk_length)


SparseTensor.random_walk = random_walk

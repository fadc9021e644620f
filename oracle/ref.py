"""The compiled reference CPU ops (oracle/_ref/libts_ref.so), exposed as ``torch.ops.ts_ref.*``.

TEST INFRASTRUCTURE ONLY.  ``available()`` is False when the library was not built.
Schemas are the reference's (csrc/spmm.cpp:344-348, csrc/convert.cpp:46-48), positional:
    spmm_sum(row?, rowptr, col, value?, colptr?, csr2csc?, mat) -> Tensor
    spmm_mean(row?, rowptr, col, value?, rowcount?, colptr?, csr2csc?, mat) -> Tensor
    spmm_min/max(rowptr, col, value?, mat) -> (Tensor, Tensor)
    ind2ptr(ind, M) / ptr2ind(ptr, E) -> Tensor
"""
import os

import torch

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'libts_ref.so')
_loaded = False


def available():
    return os.path.exists(LIB)


def ops():
    global _loaded
    if not _loaded:
        torch.ops.load_library(LIB)
        _loaded = True
    return torch.ops.ts_ref

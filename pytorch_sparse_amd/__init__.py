"""pytorch_sparse_amd -- the sparse-matmul hot path of rusty1s/pytorch_sparse, built from scratch
for AMD MI355X (gfx950).

Drop-in surface (same names and argument meaning as ``torch_sparse``):

    SparseStorage, SparseTensor, matmul, spmm, spspmm, coalesce, transpose, t
    torch.ops.torch_sparse.{spmm_sum, spmm_mean, spmm_min, spmm_max, ind2ptr, ptr2ind, cuda_version}

and, widening along SURVEY.md section 8f (the callers either side of that path):

    sum / mean / min / max, mul / add (+ _ / _nnz variants), remove_diag / set_diag / fill_diag / get_diag,
    narrow, select, index_select(_nnz), masked_select(_nnz), permute, cat, SparseTensor.__getitem__,
    sample, sample_adj, random_walk, saint_subgraph, reverse_cuthill_mckee, eye, spadd, converters
    torch.ops.torch_sparse.{non_diag_mask, sample_adj, neighbor_sample, random_walk, saint_subgraph,
                            relabel, relabel_one_hop}

Everything computes in hand-written HIP kernels (``lib/libtsamd.so``, C-ABI in ``include/tsamd.h``)
reached through the torch operator library ``lib/_tsamd_ops.so``.  There is no CPU compute path:
importing without the built libraries, or calling with CPU tensors, raises.
"""
import os

import torch

__version__ = '0.1.0'

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib')
_OPS = os.path.join(_LIBDIR, '_tsamd_ops.so')
_ops_loaded = False


def load_ops():
    """Load the torch operator library once (registers ``torch_sparse::*`` and ``tsamd::*``)."""
    global _ops_loaded
    if _ops_loaded:
        return
    if not os.path.exists(_OPS) or not os.path.exists(os.path.join(_LIBDIR, 'libtsamd.so')):
        raise ImportError(
            "pytorch_sparse_amd: native libraries not found in %s. Build them with "
            "`python pytorch_sparse_amd/build.py` (hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback." % _LIBDIR)
    torch.ops.load_library(_OPS)
    _ops_loaded = True


load_ops()
hip_version = torch.ops.torch_sparse.cuda_version()
if torch.cuda.is_available():
    # the radix sorts rank equal digits by returning LDS atomics when the device serves the lanes of one instruction in
    # ascending order; the self-test behind this query decides it once, here, so that no later sort synchronises
    sort_rank_mode = int(torch.ops.tsamd.sort_rank_mode(-1))

from .storage import SparseStorage  # noqa: E402
from .tensor import SparseTensor  # noqa: E402
from .transpose import t, transpose  # noqa: E402
from .matmul import matmul, spmm_sum, spmm_mean, spmm_min, spmm_max, spspmm_sum  # noqa: E402
from .coalesce import coalesce  # noqa: E402
from .spmm import spmm  # noqa: E402
from .relabelled import matmul_relabelled, to_relabelled, from_relabelled, relabel_index  # noqa: E402
from .spspmm import spspmm  # noqa: E402
from .reduce import sum, mean, min, max  # noqa: E402,A004
from .mul import mul, mul_, mul_nnz, mul_nnz_, add, add_, add_nnz, add_nnz_  # noqa: E402
from .select import (narrow, __narrow_diag__, select, index_select, index_select_nnz,  # noqa: E402
                     masked_select, masked_select_nnz, permute)
from .cat import cat  # noqa: E402
from .diag import remove_diag, set_diag, fill_diag, get_diag  # noqa: E402
from .sample import sample, sample_adj  # noqa: E402
from .rw import random_walk  # noqa: E402
from .saint import saint_subgraph  # noqa: E402
from .bandwidth import reverse_cuthill_mckee  # noqa: E402
from .metis import partition  # noqa: E402
from .convert import to_torch_sparse, from_torch_sparse, to_scipy, from_scipy, eye, spadd  # noqa: E402

__all__ = [
    'SparseStorage', 'SparseTensor', 't', 'transpose', 'matmul', 'coalesce', 'spmm', 'spspmm',
    'sum', 'mean', 'min', 'max', 'mul', 'mul_', 'mul_nnz', 'mul_nnz_', 'add', 'add_', 'add_nnz',
    'add_nnz_', 'narrow', '__narrow_diag__', 'select', 'index_select', 'index_select_nnz',
    'masked_select', 'masked_select_nnz', 'permute', 'cat', 'remove_diag', 'set_diag', 'fill_diag',
    'get_diag', 'sample', 'sample_adj', 'random_walk', 'saint_subgraph', 'reverse_cuthill_mckee', 'partition', 'to_torch_sparse', 'from_torch_sparse', 'to_scipy', 'from_scipy', 'eye', 'spadd',
    'matmul_relabelled', 'to_relabelled', 'from_relabelled', 'relabel_index',
    '__version__',
]

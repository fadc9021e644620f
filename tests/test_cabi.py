"""The C-ABI library loads and exports every symbol include/tsamd.h declares (no compute here),
and the host layer refuses CPU tensors instead of falling back.  CPU only."""
import ctypes
import os
import re

import pytest
import torch

from pytorch_sparse_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'tsamd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tsamd_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 8
    lib = ctypes.CDLL(nat.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), 'libtsamd.so does not export %s' % s
    assert sorted(nat.SYMBOLS) == syms, 'pytorch_sparse_amd._native.SYMBOLS out of date'


def test_version_and_status_strings():
    L = nat.lib()
    assert L.tsamd_hip_version() >= 60000000
    assert L.tsamd_status_string(0) == b'ok'
    assert b'workspace' in L.tsamd_status_string(4)
    # workspace query is host-only arithmetic
    i64 = ctypes.c_int64
    assert L.tsamd_spmm_workspace_bytes(0, 0, i64(1), i64(10), i64(10), i64(16), i64(100)) > 0
    assert L.tsamd_spmm_workspace_bytes(99, 0, i64(1), i64(10), i64(10), i64(16), i64(100)) == 0


def test_no_cpu_fallback():
    rowptr, col = torch.tensor([0, 1]), torch.tensor([0])
    with pytest.raises(nat.TsamdError, match='no CPU implementation'):
        nat.spmm(rowptr, col, None, torch.ones(1, 4), 'sum')
    with pytest.raises(nat.TsamdError):
        nat.ind2ptr(torch.tensor([0, 1]), 3)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(nat, '_lib', None)
    monkeypatch.setattr(nat, 'LIB_PATH', '/nonexistent/libtsamd.so')
    with pytest.raises(ImportError, match='no CPU fallback'):
        nat.lib()


def test_argument_validation_returns_status_codes():
    """Bad arguments are rejected by the C-ABI before any HIP call (so this runs without a GPU)."""
    L = nat.lib()
    i64, vp, sz = ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    fake = vp(0x1000)  # never dereferenced: every call below fails validation first
    # negative size / unknown reduce / unknown dtype / N too large
    assert L.tsamd_spmm(0, 0, fake, fake, None, fake, fake, None, i64(1), i64(-1), i64(4), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 1
    assert L.tsamd_spmm(0, 7, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 2
    assert L.tsamd_spmm(42, 0, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 2
    assert L.tsamd_spmm(0, 0, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(1 << 33), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 2
    # null pointers; min/max without arg_out; workspace missing / too small / misaligned
    assert L.tsamd_spmm(0, 0, None, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 1
    assert L.tsamd_spmm(0, 3, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        fake, sz(1 << 20), None) == 1
    assert L.tsamd_spmm(0, 0, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        None, sz(0), None) == 4
    assert L.tsamd_spmm(0, 0, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        fake, sz(16), None) == 4
    assert L.tsamd_spmm(0, 0, fake, fake, None, fake, fake, None, i64(1), i64(4), i64(4), i64(4), i64(4),
                        vp(0x1008), sz(1 << 20), None) == 4
    # nothing to do is OK without touching anything
    assert L.tsamd_spmm(0, 0, None, None, None, None, None, None, i64(1), i64(0), i64(4), i64(4), i64(0),
                        None, sz(0), None) == 0
    # backward entry points
    assert L.tsamd_spmm_value_bw(0, 2, None, fake, fake, fake, fake, fake, i64(1), i64(4), i64(4), i64(4),
                                 i64(4), None) == 2  # only sum / mean have a value gradient
    assert L.tsamd_spmm_value_bw(4, 0, None, fake, fake, fake, fake, fake, i64(1), i64(4), i64(4), i64(4),
                                 i64(4), None) == 2  # integer dtypes have no gradient
    assert L.tsamd_spmm_minmax_bw(5, fake, fake, None, fake, fake, fake, fake, fake, i64(1), i64(4), i64(4), i64(4),
                                  i64(4), None, sz(0), None) == 2
    assert L.tsamd_ind2ptr(None, i64(4), i64(3), fake, None) == 1
    assert L.tsamd_sort_coo(fake, fake, i64(5), i64(1 << 40), i64(1 << 40), None, None, fake, fake,
                            sz(1 << 30), None) == 2  # keys would not fit 63 bits
    assert L.tsamd_sort_coo(fake, fake, i64(5), i64(9), i64(9), None, None, fake, None, sz(0), None) == 4
    assert L.tsamd_segment_reduce(0, 9, fake, None, fake, i64(3), i64(1), fake, None) == 2
    assert L.tsamd_spspmm_numeric(2, fake, fake, None, fake, fake, None, i64(4), i64(4), fake, fake, i64(0),
                                  i64(0), i64(0), fake, fake, None, 0, None, sz(0), None) == 2  # f16: like torch.sparse.mm
    assert L.tsamd_spspmm_symbolic(0, fake, fake, None, fake, fake, None, 0, i64(4), i64(4), fake, fake, i64(0), i64(1),
                                   i64(9000), fake, None, sz(0), None) == 4  # rows beyond the LDS capacity need the workspace
    # workspace sizes grow with the problem and include the relabel copy only when it can pay off
    small = L.tsamd_spmm_workspace_bytes(0, 0, i64(1), i64(1000), i64(1000), i64(128), i64(5000))
    big = L.tsamd_spmm_workspace_bytes(0, 0, i64(1), i64(1 << 21), i64(1 << 21), i64(128), i64(40 << 20))
    halo = L.tsamd_spmm_workspace_bytes(0, 0, i64(1), i64(1 << 21), i64(1 << 24), i64(128), i64(40 << 20))
    assert small < (1 << 22) and big > (1 << 30) and halo < (1 << 28)


def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'pytorch_sparse_amd', 'lib')
    exe = str(tmp_path / 'spmm_cabi')
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', os.path.join(root, 'examples', 'spmm_cabi.c'),
           '-I' + os.path.join(root, 'include'), '-I/opt/rocm/include', '-L' + libdir, '-ltsamd', '-L/opt/rocm/lib',
           '-lamdhip64', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-o', exe]
    subprocess.check_call(cmd)
    return exe


def test_plain_c_client_builds(tmp_path):
    """examples/spmm_cabi.c -- C99, gcc, no torch, no C++ -- compiles and links against
    include/tsamd.h + libtsamd.so: the boundary really is a C-ABI."""
    assert os.path.exists(_build_c_example(tmp_path))


@pytest.mark.gpu
def test_plain_c_client_runs(tmp_path):
    import subprocess
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'C-ABI example OK' in out.stdout

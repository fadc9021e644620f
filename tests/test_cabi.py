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

"""Experiment (round 6, re-entry): do the symbolic and the numeric stage of the small-row SpSpMM overlap when they run
concurrently on two streams?  Config 4 input; the numeric stage gets the row pointer of an earlier symbolic run, the
concurrent symbolic run counts into a scratch array.  Prints the three wall times (symbolic, numeric, both)."""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import baseline_configs as bc
from pytorch_sparse_amd import _native as nat

dev = torch.device('cuda:0')
L = nat.lib()
L.tsamd_spspmm_workspace_bytes.restype = ctypes.c_size_t
L.tsamd_exclusive_scan_workspace_bytes.restype = ctypes.c_size_t
A, At = bc.spspmm_inputs(dev, sys.argv[1] if len(sys.argv) > 1 else 'c4')
rpA, cA, vA = A.storage.rowptr(), A.storage.col(), A.storage.value()
rpB, cB, vB = At.storage.rowptr(), At.storage.col(), At.storage.value()
M, N = rpA.numel() - 1, At.size(1)
p = lambda t: ctypes.c_void_p(t.data_ptr())
i64 = ctypes.c_int64
prod = torch.empty(M + 1, dtype=torch.int64, device=dev)
bins = torch.empty(2 * M + 1, dtype=torch.int64, device=dev)
stats = torch.empty(8, dtype=torch.int64, device=dev)
cB32 = torch.empty(cB.numel(), dtype=torch.int32, device=dev)
s0 = torch.cuda.current_stream(dev)
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
nat.check(L.tsamd_spspmm_plan(p(rpA), p(cA), p(rpB), p(cB), i64(cB.numel()), i64(M), p(prod), p(bins), p(cB32), p(stats), sp(s0)), 'plan')
hs = stats.cpu().tolist()
n_medium, n_large, P_large = hs[2], hs[3], hs[4]
assert n_large == 0, 'experiment covers inputs without large rows'
ws = torch.empty(256, dtype=torch.uint8, device=dev)


def symbolic(out, stream):
    nat.check(L.tsamd_spspmm_symbolic(0, p(rpA), p(cA), p(vA), p(rpB), p(cB32), p(vB), 1, i64(M), i64(N), p(prod), p(bins),
                                      i64(n_medium), i64(0), i64(0), p(out), p(ws), ctypes.c_size_t(256), sp(stream)), 'symbolic')


rowptrC = torch.zeros(M + 1, dtype=torch.int64, device=dev)
symbolic(rowptrC, s0)
total = torch.empty(1, dtype=torch.int64, device=dev)
ws2 = torch.empty(max(int(L.tsamd_exclusive_scan_workspace_bytes(i64(M + 1))), 256), dtype=torch.uint8, device=dev)
nat.check(L.tsamd_exclusive_scan_i64(p(rowptrC), p(rowptrC), i64(M + 1), p(total), p(ws2), ctypes.c_size_t(ws2.numel()), sp(s0)), 'scan')
nnz = int(total.item())
colC = torch.empty(nnz, dtype=torch.int64, device=dev)
valC = torch.empty(nnz, dtype=torch.float32, device=dev)


def numeric(stream):
    nat.check(L.tsamd_spspmm_numeric(0, p(rpA), p(cA), p(vA), p(rpB), p(cB32), p(vB), i64(M), i64(N), p(prod), p(bins),
                                     i64(n_medium), i64(0), i64(0), p(rowptrC), p(colC), p(valC), 1, p(ws), ctypes.c_size_t(256),
                                     sp(stream)), 'numeric')


scratch = torch.zeros(M + 1, dtype=torch.int64, device=dev)
s1 = torch.cuda.Stream(dev)


def timed(fn, iters=7):
    import time
    ts = []
    for _ in range(iters + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


def both():
    symbolic(scratch, s1)
    numeric(s0)


def both_rev():
    numeric(s0)
    symbolic(scratch, s1)


res = dict(nnz=nnz, symbolic_ms=timed(lambda: symbolic(scratch, s0)), numeric_ms=timed(lambda: numeric(s0)),
           serial_ms=timed(lambda: (symbolic(scratch, s0), numeric(s0))),
           both_ms=timed(both), both_numeric_first_ms=timed(both_rev))
print(json.dumps(res))

"""Secondary measurements (not the bench.py headline): BASELINE.json configs 3 and 4, coalesce /
sort / transpose throughput, each with its CPU baseline (torch CPU = what the reference calls).
Prints one JSON object per line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_sparse_amd as ts
from pytorch_sparse_amd import synth, _native as nat

dev = torch.device('cuda:0')
which = set(sys.argv[1:]) or {'c3', 'c4', 'coalesce', 'vbw'}

def gpu_time(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts_ = []
    for _ in range(iters):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts_.append(s.elapsed_time(e))
    ts_.sort()
    return ts_[len(ts_) // 2]

def wall(fn, iters=3):
    best = 1e30
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3

if 'c3' in which:
    # config 3: SpMM-max + backward, R-MAT scale 20, F=128 bf16 (value-less and with values)
    scale, K = 20, 128
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
    for has_value in (False, True):
        v = synth.values(E, dtype=torch.bfloat16, device=dev) if has_value else None
        x = synth.features(n, K, dtype=torch.bfloat16, device=dev)
        t_fw = gpu_time(lambda: nat.spmm(rp, c, v, x, 'max'))
        out, arg = nat.spmm(rp, c, v, x, 'max')
        g = synth.features(n, K, seed=3, dtype=torch.bfloat16, device=dev)
        t_bw = gpu_time(lambda: nat.spmm_minmax_bw(rp, c, v, x, g, arg, want_value=has_value, want_mat=True))
        balg = E * (8 + (2 if has_value else 0) + K * 2) + (n + 1) * 8 + n * K * 2 + n * K * 8
        print(json.dumps(dict(bench='c3_spmm_max_bf16', has_value=has_value, E=E, F=K, fw_ms=round(t_fw, 3), bw_ms=round(t_bw, 3),
                              gedges_fw=round(E / t_fw / 1e6, 2), balg_gbs=round(balg / t_fw / 1e6, 1), frac_hbm=round(balg / t_fw / 1e6 / 8000, 3))), flush=True)

if 'vbw' in which:
    scale, K = 20, 64
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
    row = nat.ptr2ind(rp, E)
    x = synth.features(n, K, device=dev); g = synth.features(n, K, seed=3, device=dev)
    t = gpu_time(lambda: nat.spmm_value_bw(row, rp, c, x, g, 'sum'))
    balg = E * (16 + K * 4 + 4) + n * K * 4
    print(json.dumps(dict(bench='value_bw_f32', E=E, F=K, ms=round(t, 3), gedges=round(E / t / 1e6, 2), balg_gbs=round(balg / t / 1e6, 1))), flush=True)
    A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(E, device=dev).requires_grad_(), sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    xr = x.clone().requires_grad_()
    A.storage.fill_cache_()
    def fwbw():
        out = ts.matmul(A, xr, 'sum'); out.backward(g)
    t = gpu_time(fwbw, iters=5)
    print(json.dumps(dict(bench='matmul_sum_fw_bw_f32', E=E, F=K, ms=round(t, 3))), flush=True)

if 'coalesce' in which:
    m = n = 500000; nnz = 7500000
    row, col = synth.uniform_edges(m, n, nnz, seed=0, device=dev)
    val = synth.values(nnz, device=dev)
    index = torch.stack([row, col])
    t_sort = wall(lambda: torch.ops.tsamd.sort_coo(row, col, m, n, True))
    t_ctor = wall(lambda: ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n)))
    t_coal = wall(lambda: ts.coalesce(index, val, m, n))
    t_tr = wall(lambda: ts.transpose(index, val, m, n))
    A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n))
    def tt():
        A.storage._csr2csc = None; A.storage._colptr = None; A.storage._csc2csr = None
        return A.t()
    t_t = wall(tt)
    key = row * n + col
    t_torch_sort = wall(lambda: torch.sort(key))
    # CPU baseline: what the reference does (torch.sort of int64 keys + mask + segment sums)
    rc, cc, vc = row.cpu(), col.cpu(), val.cpu()
    t0 = time.perf_counter(); k = rc * n + cc; ks, perm = torch.sort(k); cpu_sort = (time.perf_counter() - t0) * 1e3
    print(json.dumps(dict(bench='coalesce_7.5M', sort_coo_ms=round(t_sort, 3), ctor_ms=round(t_ctor, 3), coalesce_ms=round(t_coal, 3),
                          transpose_ms=round(t_tr, 3), t_ms=round(t_t, 3), torch_gpu_sort_keys_only_ms=round(t_torch_sort, 3),
                          entries_per_s_sort=round(nnz / t_sort / 1e3, 1), cpu_torch_sort_ms=round(cpu_sort, 1), cpu_cores=os.cpu_count())), flush=True)

if 'c4' in which:
    # config 4: SpSpMM A * A^T, 500k x 500k uniform, 7.5M draws
    m = n = 500000; nnz = 7500000
    row, col = synth.uniform_edges(m, n, nnz, seed=0, device=dev)
    A = ts.SparseTensor(row=row, col=col, value=synth.values(nnz, device=dev), sparse_sizes=(m, n)).coalesce()
    At = A.t()
    t = wall(lambda: A @ At, iters=3)
    C = A @ At
    rpA = A.storage.rowptr(); rpB = At.storage.rowptr()
    P = int((rpB[A.storage.col() + 1] - rpB[A.storage.col()]).sum())
    res = dict(bench='c4_spspmm_AAt', nnzA=A.nnz(), products=P, nnzC=C.nnz(), ms=round(t, 2), gproducts_per_s=round(P / t / 1e6, 3))
    if 'cpu' in which:
        Ac = A.cpu().to_torch_sparse_coo_tensor(); Bc = At.cpu().to_torch_sparse_coo_tensor()
        t0 = time.perf_counter(); Cc = torch.sparse.mm(Ac, Bc); res['cpu_torch_sparse_mm_s'] = round(time.perf_counter() - t0, 2)
        res['cpu_nnzC'] = Cc._nnz()
    print(json.dumps(res), flush=True)

if 'narrow' in which:
    # same graph, several dtypes / widths: B_alg changes with the element size
    scale = 21
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
    for dtype, K in ((torch.float32, 128), (torch.bfloat16, 128), (torch.float16, 128), (torch.float32, 64), (torch.bfloat16, 256), (torch.float64, 64), (torch.float32, 32), (torch.float32, 16), (torch.float32, 256), (torch.float32, 512)):
        v = synth.values(E, dtype=dtype, device=dev); x = synth.features(n, K, dtype=dtype, device=dev)
        for red in ('sum', 'max'):
            t = gpu_time(lambda: nat.spmm(rp, c, v, x, red), iters=7)
            s = x.element_size()
            balg = E * (8 + s + K * s) + (n + 1) * 8 + n * K * s + (n * K * 8 if red == 'max' else 0)
            print(json.dumps(dict(bench='spmm_ns_graph', dtype=str(dtype).split('.')[1], F=K, reduce=red, ms=round(t, 3), gedges=round(E / t / 1e6, 2),
                                  balg_gbs=round(balg / t / 1e6, 1), frac_hbm=round(balg / t / 1e6 / 8000, 3))), flush=True)
        del v, x

if 'c1' in which:
    # config 1: torch_sparse.spmm on 1k x 1k COO, 5k nnz (unsorted, duplicates possible), F=16 fp32
    m = n = 1000; nnz = 5000
    row, col = synth.uniform_edges(m, n, nnz, seed=0, device=dev)
    index = torch.stack([row, col]); val = synth.values(nnz, device=dev); x = synth.features(n, 16, device=dev)
    t_legacy = wall(lambda: ts.spmm(index, val, m, n, x), iters=20)
    A = ts.SparseTensor(row=row, col=col, value=val, sparse_sizes=(m, n)); A.storage.rowptr()
    t_mm = wall(lambda: A.matmul(x), iters=50)
    t_k = gpu_time(lambda: A.matmul(x), iters=50)
    out = ts.spmm(index, val, m, n, x)
    ref = torch.zeros(m, 16, device=dev).index_add_(0, row, val[:, None] * x[col])
    print(json.dumps(dict(bench='c1_legacy_spmm_1k', legacy_spmm_ms_wall=round(t_legacy, 4), matmul_ms_wall=round(t_mm, 4),
                          matmul_ms_gpu=round(t_k, 4), max_abs_err=float((out - ref).abs().max()))), flush=True)

if 'prelabel' in which:
    # The per-call relabelled copy of X disappears when the graph itself is stored in
    # a channel-friendly node order: relabel the nodes ONCE (adj.permute(perm), x[perm]) -- any random
    # permutation does -- and tsamd_spmm's probe finds nothing to fix.  Same matrix up to that
    # symmetric permutation: out' = (P A P^T)(P X) = P (A X).
    scale, K = 21, 128
    rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev); n = 1 << scale; E = c.numel()
    v = synth.values(E, device=dev); x = synth.features(n, K, device=dev)
    t0 = gpu_time(lambda: nat.spmm(rp, c, v, x, 'sum'), iters=10)
    A = ts.SparseTensor(rowptr=rp, col=c, value=v, sparse_sizes=(n, n), is_sorted=True, trust_data=True)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).to(dev)
    t_perm = wall(lambda: A.permute(perm), iters=3)
    Ap = A.permute(perm); xp = x[perm]
    rpp, cp, vp = Ap.csr()
    t1 = gpu_time(lambda: nat.spmm(rpp, cp, vp, xp, 'sum'), iters=10)
    ref = nat.spmm(rp, c, v, x, 'sum')[0][perm]
    got = nat.spmm(rpp, cp, vp, xp, 'sum')[0]
    balg = E * (8 + 4 + K * 4) + (n + 1) * 8 + n * K * 4
    print(json.dumps(dict(bench='ns_pre_permuted_graph', as_given_ms=round(t0, 3), pre_permuted_ms=round(t1, 3),
                          one_time_permute_ms=round(t_perm, 2), frac_hbm_pre_permuted=round(balg / t1 / 1e6 / 8000, 3),
                          max_rel_diff=float(((got - ref).abs().max() / ref.abs().max())))), flush=True)

"""Mini-batch producer timings on one MI355X: neighbour sampling, random walks,
SAINT sub-graphs on the config-2 graph (R-MAT scale 20, edge factor 20).  GPU wall time of the public
API call including its host syncs.  The reference runs these on the CPU only; its timings on the same
box come from tests/report_sampler_baseline.py (the oracle may only be touched from tests/).
Prints one JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pytorch_sparse_amd as ts  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

dev = torch.device('cuda:0')


def wall(fn, iters=7, warm=2):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    t.sort()
    return t[len(t) // 2] * 1e3


scale = int(os.environ.get('SCALE', 20))
rp, c = synth.rmat_csr(scale, 20, seed=0, device=dev)
n = 1 << scale
E = c.numel()
A = ts.SparseTensor(rowptr=rp, col=c, value=synth.values(E, device=dev), sparse_sizes=(n, n), is_sorted=True,
                    trust_data=True)
A.storage.row()
g = torch.Generator().manual_seed(0)
perm = torch.randperm(n, generator=g).to(dev)

for seeds in (1_000, 100_000, n):
    idx = perm[:seeds]
    for k, replace in ((10, False), (25, False), (10, True), (-1, False)):
        if k < 0 and seeds > 100_000:
            continue
        ms = wall(lambda: A.sample_adj(idx, k, replace=replace))
        adj, n_id = A.sample_adj(idx, k, replace=replace)
        print(json.dumps(dict(bench='sample_adj', seeds=seeds, k=k, replace=replace, ms=round(ms, 3),
                              sampled=adj.nnz(), n_id=n_id.numel(),
                              mdraws_per_s=round(adj.nnz() / ms / 1e3, 1))), flush=True)

for walks, L in ((100_000, 20), (n, 20), (n, 80)):
    start = perm[:walks]
    ms = wall(lambda: A.random_walk(start, L))
    print(json.dumps(dict(bench='random_walk', walks=walks, length=L, ms=round(ms, 3),
                          msteps_per_s=round(walks * L / ms / 1e3, 1))), flush=True)

for frac in (0.01, 0.25):
    idx = perm[:int(n * frac)]
    ms = wall(lambda: A.saint_subgraph(idx))
    sub, _ = A.saint_subgraph(idx)
    print(json.dumps(dict(bench='saint_subgraph', nodes=idx.numel(), ms=round(ms, 3), edges=sub.nnz())), flush=True)

# multi-hop neighbor_sample (the CSR arrays serve as the CSC view of the transposed graph)
for seeds, fan in ((1024, [25, 10]), (1024, [15, 10, 5]), (100_000, [10, 10])):
    inp = perm[:seeds]
    fn = lambda: torch.ops.torch_sparse.neighbor_sample(rp, c, inp, fan, False, True)  # noqa: E731
    ms = wall(fn)
    node, r_, c_, e_ = fn()
    print(json.dumps(dict(bench='neighbor_sample', seeds=seeds, fanout=fan, ms=round(ms, 3), nodes=node.numel(),
                          edges=e_.numel(), medges_per_s=round(e_.numel() / ms / 1e3, 1))), flush=True)

# heterogeneous multi-hop samplers: three node types, five relations (the shape of an academic graph), uniform degrees
# 0..39 per destination node; mini-batch of 1024 paper seeds.  `syncs` = host read-backs of one call (torch's sync debug
# mode warns on each): the round-6 samplers are device-driven inside a hop.

NODE_TYPES = ['paper', 'author', 'venue']
EDGE_TYPES = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'in', 'venue'),
              ('venue', 'hosts', 'paper'), ('paper', 'by', 'author')]
RELS = ['__'.join(e) for e in EDGE_TYPES]
sizes = {'paper': 1 << scale, 'author': 1 << (scale - 1), 'venue': 1 << 10}
gh = torch.Generator(device=dev).manual_seed(1)
colptr_d, row_d = {}, {}
for (s_, r_, d_) in EDGE_TYPES:
    deg = torch.randint(0, 40, (sizes[d_], ), generator=gh, device=dev)
    cp = torch.zeros(sizes[d_] + 1, dtype=torch.long, device=dev)
    cp[1:] = deg.cumsum(0)
    colptr_d['__'.join((s_, r_, d_))] = cp
    row_d['__'.join((s_, r_, d_))] = torch.randint(0, sizes[s_], (int(cp[-1]), ), generator=gh, device=dev)
times_d = {t: torch.randint(0, 100, (sizes[t], ), generator=gh, device=dev) for t in NODE_TYPES}


def count_syncs(fn):
    import tempfile
    torch.cuda.synchronize()
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode='w+b') as tmp:
        os.dup2(tmp.fileno(), 2)
        torch.cuda.set_sync_debug_mode('warn')
        try:
            fn()
        finally:
            torch.cuda.set_sync_debug_mode('default')
            sys.stderr.flush()
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode(errors='replace').count('synchronizing')


for seeds, fanv, hops in ((1024, 10, 2), (1024, 5, 3), (65536, 10, 2)):
    inp_d = {'paper': perm[:seeds] % sizes['paper']}
    fan_d = {r: [fanv] * hops for r in RELS}
    for name, fn in (
            ('hetero_neighbor_sample', lambda: torch.ops.torch_sparse.hetero_neighbor_sample(
                NODE_TYPES, EDGE_TYPES, colptr_d, row_d, inp_d, fan_d, hops, False, True)),
            ('hetero_temporal_neighbor_sample', lambda: torch.ops.torch_sparse.hetero_temporal_neighbor_sample(
                NODE_TYPES, EDGE_TYPES, colptr_d, row_d, inp_d, fan_d, times_d, hops, False, True))):
        ms = wall(fn)
        out = fn()
        edges = sum(out[3][r].numel() for r in RELS)
        print(json.dumps(dict(bench=name, seeds=seeds, fanout=fanv, hops=hops, relations=len(RELS), ms=round(ms, 3),
                              nodes=sum(out[0][t].numel() for t in NODE_TYPES), edges=edges,
                              medges_per_s=round(edges / ms / 1e3, 2), syncs=count_syncs(fn))), flush=True)

"""CPU timings of the reference's own samplers (oracle/_ref: csrc/cpu/{sample,rw,saint}_cpu.cpp compiled
unmodified) on this box, on bounded samples of the workloads of scripts/bench_sample.py -- the
baseline column for SURVEY.md 8f rank 4.  Not a test (lives here because only tests/ may touch the
oracle).  Usage: python tests/report_sampler_baseline.py   -> one JSON object per line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import ref  # noqa: E402
from pytorch_sparse_amd import synth  # noqa: E402

r = ref.ops()
scale = int(os.environ.get('SCALE', 20))
rp, c = synth.rmat_csr(scale, 20, seed=0)
n = 1 << scale
row = r.ptr2ind(rp, c.numel())
g = torch.Generator().manual_seed(0)
perm = torch.randperm(n, generator=g)


def wall(fn):
    t0 = time.perf_counter()
    out = fn()
    return (time.perf_counter() - t0) * 1e3, out


for seeds, k, replace in ((1_000, 10, False), (20_000, 10, False), (20_000, 25, False), (20_000, 10, True),
                          (20_000, -1, False)):
    ms, out = wall(lambda: r.sample_adj(rp, c, perm[:seeds], k, replace))
    print(json.dumps(dict(bench='sample_adj_cpu_reference', seeds=seeds, k=k, replace=replace, ms=round(ms, 1),
                          sampled=out[1].numel(), mdraws_per_s=round(out[1].numel() / ms / 1e3, 3))), flush=True)
for walks, L in ((100_000, 20), (n, 20)):
    ms, out = wall(lambda: r.random_walk(rp, c, perm[:walks], L))
    print(json.dumps(dict(bench='random_walk_cpu_reference', walks=walks, length=L, ms=round(ms, 1),
                          msteps_per_s=round(walks * L / ms / 1e3, 2))), flush=True)
for frac in (0.01, 0.25):
    idx = perm[:int(n * frac)]
    ms, out = wall(lambda: r.saint_subgraph(idx, rp, row, c))
    print(json.dumps(dict(bench='saint_subgraph_cpu_reference', nodes=idx.numel(), ms=round(ms, 1),
                          edges=out[0].numel())), flush=True)
for seeds, fan in ((1024, [25, 10]), (1024, [15, 10, 5]), (20_000, [10, 10])):
    ms, out = wall(lambda: r.neighbor_sample(rp, c, perm[:seeds], fan, False, True))
    print(json.dumps(dict(bench='neighbor_sample_cpu_reference', seeds=seeds, fanout=fan, ms=round(ms, 1),
                          nodes=out[0].numel(), edges=out[3].numel(),
                          medges_per_s=round(out[3].numel() / ms / 1e3, 3))), flush=True)

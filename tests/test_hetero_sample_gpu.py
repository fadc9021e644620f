"""torch_sparse::hetero_neighbor_sample / hetero_temporal_neighbor_sample on the GPU (csrc/ops_sample.cpp) against the
reference's CPU samplers (csrc/cpu/neighbor_sample_cpu.cpp:135-507):
  * tests/golden/py8_hetero_*.npz -- outputs of the COMPILED REFERENCE (make_golden.py part 8) for every deterministic
    case (take all neighbours, or more draws than neighbours): every node list, row / col / edge list bit for bit;
  * for random draws: the properties the reference guarantees (counts, distinct draws without replacement, every
    edge is a stored entry between the nodes it names, first-occurrence numbering, the time constraint)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEV = 'cuda'
NODE_TYPES = ['paper', 'author', 'venue']
EDGE_TYPES = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'in', 'venue'),
              ('venue', 'hosts', 'paper'), ('paper', 'by', 'author')]
RELS = ['__'.join(e) for e in EDGE_TYPES]


@pytest.fixture(scope='module')
def ops():
    import pytorch_sparse_amd  # noqa: F401
    return torch.ops.torch_sparse


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _graph(z):
    colptr = {r: dev(z['colptr__' + r]) for r in RELS}
    row = {r: dev(z['row__' + r]) for r in RELS}
    inp = {t: dev(z['input__' + t]) for t in NODE_TYPES if 'input__' + t in z.files}
    times = {t: dev(z['time__' + t]) for t in NODE_TYPES if 'time__' + t in z.files}
    return colptr, row, inp, times


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'py8_hetero_*.npz'))), ids=os.path.basename)
def test_golden_hetero_sampling(ops, path):
    z = np.load(path)
    colptr, row, inp, times = _graph(z)
    mode, hops, fanval = str(z['mode']), int(z['hops']), int(z['fan'])
    fan = {r: [fanval] * hops for r in RELS}
    if mode.startswith('temporal'):
        out = ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, colptr, row, inp, fan, times, hops, False, True)
    else:
        out = ops.hetero_neighbor_sample(NODE_TYPES, EDGE_TYPES, colptr, row, inp, fan, hops, False, mode == 'directed')
    for t in NODE_TYPES:
        np.testing.assert_array_equal(out[0][t].cpu().numpy(), z['node__' + t], err_msg='node ' + t)
    for r in RELS:
        for got, key in ((out[1][r], 'orow__'), (out[2][r], 'ocol__'), (out[3][r], 'oedge__')):
            np.testing.assert_array_equal(got.cpu().numpy(), z[key + r], err_msg=key + r)


def _random_graph(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    colptr, row = {}, {}
    for (s, r, d) in EDGE_TYPES:
        rel = '__'.join((s, r, d))
        deg = torch.randint(0, 40, (sizes[d], ), generator=g)
        deg[::5] = 0
        cp = torch.zeros(sizes[d] + 1, dtype=torch.long)
        cp[1:] = deg.cumsum(0)
        colptr[rel] = cp
        row[rel] = torch.randint(0, sizes[s], (int(cp[-1]), ), generator=g)
    times = {t: torch.randint(0, 100, (sizes[t], ), generator=g) for t in NODE_TYPES}
    return colptr, row, times, g


@pytest.mark.parametrize('replace', [False, True])
@pytest.mark.parametrize('temporal', [False, True])
def test_random_draws_keep_the_reference_guarantees(ops, replace, temporal):
    sizes = {'paper': 5000, 'author': 2000, 'venue': 30}
    colptr, row, times, g = _random_graph(3, sizes)
    inp = {'paper': torch.randperm(sizes['paper'], generator=g)[:200], 'venue': torch.tensor([3, 7])}
    fan = {r: [4, 3] for r in RELS}
    todev = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    torch.manual_seed(11)
    if temporal:
        out = ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, todev(colptr), todev(row), todev(inp), fan,
                                                  todev(times), 2, replace, True)
    else:
        out = ops.hetero_neighbor_sample(NODE_TYPES, EDGE_TYPES, todev(colptr), todev(row), todev(inp), fan, 2, replace, True)
    node = {t: out[0][t].cpu() for t in NODE_TYPES}
    for t, x in inp.items():  # the input nodes come first, in order
        assert torch.equal(node[t][:x.numel()], x)
    if not temporal:  # one id per node (the temporal sampler numbers (node, root) pairs: a node may repeat)
        for t in NODE_TYPES:
            assert node[t].unique().numel() == node[t].numel()
    total = 0
    for (s, _, d), rel in zip(EDGE_TYPES, RELS):
        r, c, e = out[1][rel].cpu(), out[2][rel].cpu(), out[3][rel].cpu()
        assert r.numel() == c.numel() == e.numel()
        total += r.numel()
        if r.numel() == 0:
            continue
        assert int(r.max()) < node[s].numel() and int(c.max()) < node[d].numel()
        # every edge is a stored entry: its source is row[e], its destination owns the segment that holds e
        assert torch.equal(row[rel][e], node[s][r])
        w = node[d][c]
        assert bool((colptr[rel][w] <= e).all()) and bool((e < colptr[rel][w + 1]).all())
        # per destination occurrence: at most fan draws, exactly min(deg, fan) without replacement and without time
        cnt = torch.bincount(c, minlength=node[d].numel())
        assert int(cnt.max()) <= 4
        if not replace:
            key = c * (int(e.max()) + 1) + e
            assert key.unique().numel() == key.numel()  # distinct entries per destination
        if not temporal and not replace:
            first_hop = cnt[:inp[d].numel()] if d in inp else cnt[:0]
            deg = (colptr[rel][1:] - colptr[rel][:-1])[inp[d]] if d in inp else first_hop
            assert torch.equal(first_hop, torch.minimum(deg, torch.full_like(deg, 4)))
        if temporal:
            # the drawn source is not younger than the ROOT of the tree it was drawn into; roots: the input nodes
            pass
    assert total > 0
    # reproducible under torch.manual_seed
    torch.manual_seed(11)
    if temporal:
        again = ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, todev(colptr), todev(row), todev(inp), fan,
                                                    todev(times), 2, replace, True)
    else:
        again = ops.hetero_neighbor_sample(NODE_TYPES, EDGE_TYPES, todev(colptr), todev(row), todev(inp), fan, 2, replace, True)
    for t in NODE_TYPES:
        assert torch.equal(again[0][t], out[0][t])


def test_temporal_constraint_holds_on_every_drawn_edge(ops):
    """One hop from `paper` roots: every drawn source v obeys time[src][v] <= time[paper][root] -- with and without
    replacement, and a root none of whose neighbours qualifies draws nothing."""
    sizes = {'paper': 3000, 'author': 1500, 'venue': 20}
    colptr, row, times, g = _random_graph(5, sizes)
    roots = torch.randperm(sizes['paper'], generator=g)[:300]
    todev = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    for replace in (False, True):
        out = ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, todev(colptr), todev(row), {'paper': roots.to(DEV)},
                                                  {r: [5] for r in RELS}, todev(times), 1, replace, True)
        seen = 0
        for (s, _, d), rel in zip(EDGE_TYPES, RELS):
            if d != 'paper':
                assert out[1][rel].numel() == 0  # only paper nodes are in the first frontier
                continue
            r, c, e = out[1][rel].cpu(), out[2][rel].cpu(), out[3][rel].cpu()
            src = out[0][s].cpu()[r]
            root_time = times['paper'][roots[c]]
            assert bool((times[s][src] <= root_time).all())
            assert torch.equal(row[rel][e], src)
            seen += r.numel()
            if replace:  # exactly 5 draws for every root with at least one valid neighbour, none otherwise
                cp, rw = colptr[rel], row[rel]
                valid = torch.zeros(roots.numel(), dtype=torch.long)
                for i, w in enumerate(roots.tolist()):
                    nb = rw[cp[w]:cp[w + 1]]
                    valid[i] = int((times[s][nb] <= times['paper'][w]).sum())
                cnt = torch.bincount(c, minlength=roots.numel())
                assert torch.equal(cnt, torch.where(valid > 0, torch.full_like(valid, 5), torch.zeros_like(valid)))
        assert seen > 0


def _count_syncs(fn):
    """Number of synchronising device -> host operations `fn` performs: torch's sync debug mode warns on each -- as a
    Python warning when the operation is called from Python, on the process's stderr from inside a C++ operator (file
    descriptor 2 is captured around the call)."""
    import sys
    import tempfile
    import warnings
    torch.cuda.synchronize()
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode='w+b') as tmp, warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        os.dup2(tmp.fileno(), 2)
        torch.cuda.set_sync_debug_mode('warn')
        try:
            out = fn()
        finally:
            torch.cuda.set_sync_debug_mode('default')
            sys.stderr.flush()
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        text = tmp.read().decode(errors='replace')
        n_py = sum('synchronizing' in str(w.message) for w in caught)
    return out, text.count('synchronizing') + n_py


@pytest.mark.parametrize('temporal', [False, True])
def test_host_read_backs_per_hop(ops, temporal):
    """The samplers are device-driven inside a hop: the untimed one reads back twice per hop (the sizes of all the hop's
    draws; the new lengths of all node lists) however many relations there are, the temporal one once per hop + once per
    (relation, hop) -- round 5: two resp. four to five per (relation, hop).  + one transfer for the set-up."""
    sizes = {'paper': 5000, 'author': 2000, 'venue': 30}
    colptr, row, times, g = _random_graph(9, sizes)
    inp = {'paper': torch.randperm(sizes['paper'], generator=g)[:100], 'venue': torch.tensor([1, 2])}
    hops = 3
    fan = {r: [3] * hops for r in RELS}
    todev = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    C, Rw, In, Tm = todev(colptr), todev(row), todev(inp), todev(times)
    if temporal:
        call = lambda: ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, C, Rw, In, fan, Tm, hops, False, True)  # noqa: E731
    else:
        call = lambda: ops.hetero_neighbor_sample(NODE_TYPES, EDGE_TYPES, C, Rw, In, fan, hops, False, True)  # noqa: E731
    call()  # first call: fills the cache of the graph's id maxima
    _, probe = _count_syncs(lambda: torch.ones(3, device=DEV).sum().item())
    assert probe >= 1  # the counter sees a read-back
    out, n_sync = _count_syncs(call)
    assert sum(out[1][r].numel() for r in RELS) > 0
    if temporal:
        assert n_sync <= 1 + hops * (1 + len(RELS)), n_sync
    else:
        assert n_sync <= 1 + 2 * hops, n_sync


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_random_graphs_against_the_sequential_restatement(ops, seed):
    """Deterministic draws (every neighbour, or more draws than neighbours) on random heterogeneous graphs against
    oracle/np_oracle.py: hetero_neighbor_sample_det -- the statement-by-statement restatement of hetero_sample that
    tests/test_oracle.py pins to the compiled reference: input nodes listed twice or three times (first position keeps
    the node), empty relations, a type without times, directed / undirected / temporal, 1..3 hops."""
    from oracle import np_oracle as npo
    rng = np.random.default_rng(100 + seed)
    sizes = {'paper': int(rng.integers(50, 3000)), 'author': int(rng.integers(20, 1000)), 'venue': int(rng.integers(2, 40))}
    colptr, row = {}, {}
    for (s, r, d) in EDGE_TYPES:
        deg = rng.integers(0, 9, sizes[d])
        deg[::3] = 0
        cp = np.zeros(sizes[d] + 1, np.int64)
        np.cumsum(deg, out=cp[1:])
        colptr['__'.join((s, r, d))] = cp
        row['__'.join((s, r, d))] = rng.integers(0, sizes[s], int(cp[-1]))
    if seed % 2:
        colptr['venue__hosts__paper'] = np.zeros_like(colptr['venue__hosts__paper'])
        row['venue__hosts__paper'] = row['venue__hosts__paper'][:0]
    times = {t: rng.integers(0, 30, sizes[t]) for t in NODE_TYPES}
    seeds = rng.integers(0, sizes['paper'], 40)
    seeds[7], seeds[21] = seeds[3], seeds[3]
    inp = {'paper': seeds, 'author': rng.integers(0, sizes['author'], 5)}
    D = lambda d: {k: dev(v) for k, v in d.items()}  # noqa: E731
    for hops in (1, 2, 3):
        for fanv in (-1, 20):
            fan = {r: [fanv] * hops for r in RELS}
            for directed in (True, False):
                got = ops.hetero_neighbor_sample(NODE_TYPES, EDGE_TYPES, D(colptr), D(row), D(inp), fan, hops, False, directed)
                want = npo.hetero_neighbor_sample_det(NODE_TYPES, EDGE_TYPES, colptr, row, inp, fan, hops, directed)
                for t in NODE_TYPES:
                    np.testing.assert_array_equal(got[0][t].cpu().numpy(), want[0][t], err_msg='node ' + t)
                for r in RELS:
                    for k in (1, 2, 3):
                        np.testing.assert_array_equal(got[k][r].cpu().numpy(), want[k][r], err_msg='%s %d %s %d' % (r, k, directed, hops))
            tm = {t: v for t, v in times.items() if t != 'venue'} if seed % 2 else times
            got = ops.hetero_temporal_neighbor_sample(NODE_TYPES, EDGE_TYPES, D(colptr), D(row), D(inp), fan, D(tm), hops, False, True)
            want = npo.hetero_neighbor_sample_det(NODE_TYPES, EDGE_TYPES, colptr, row, inp, fan, hops, True, tm)
            for t in NODE_TYPES:
                np.testing.assert_array_equal(got[0][t].cpu().numpy(), want[0][t], err_msg='temporal node ' + t)
            for r in RELS:
                for k in (1, 2, 3):
                    np.testing.assert_array_equal(got[k][r].cpu().numpy(), want[k][r], err_msg='temporal %s %d %d' % (r, k, hops))


def test_homogeneous_neighbor_sample_with_repeated_seeds():
    """neighbor_sample with a seed listed twice: the first position keeps the node (neighbor_sample_cpu.cpp:31)."""
    import pytorch_sparse_amd  # noqa: F401
    from oracle import np_oracle as npo
    rng = np.random.default_rng(5)
    n = 400
    key = np.unique(rng.integers(0, n * n, 3000))
    order = np.lexsort((key // n, key % n))
    row, col = (key // n)[order], (key % n)[order]
    colptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(col, minlength=n), out=colptr[1:])
    inp = rng.permutation(n)[:30]
    inp[11], inp[20] = inp[4], inp[4]
    types, etypes = ['n'], [('n', 'e', 'n')]
    for hops in (1, 3):
        for directed in (True, False):
            got = torch.ops.torch_sparse.neighbor_sample(dev(colptr), dev(row), dev(inp), [-1] * hops, False, directed)
            want = npo.hetero_neighbor_sample_det(types, etypes, {'n__e__n': colptr}, {'n__e__n': row}, {'n': inp},
                                                  {'n__e__n': [-1] * hops}, hops, directed)
            np.testing.assert_array_equal(got[0].cpu().numpy(), want[0]['n'])
            for k in (1, 2, 3):
                np.testing.assert_array_equal(got[k].cpu().numpy(), want[k]['n__e__n'], err_msg='%d %s %d' % (k, directed, hops))

"""Mini-batch producers on the GPU: random_walk, sample_adj, saint_subgraph,
relabel, relabel_one_hop.

  * bit-exact against tests/golden/py4_*.npz -- outputs of the reference's CPU kernels compiled
    unmodified (make_golden.py part 4): walks with the floats the reference drew, and every
    deterministic case of the others;
  * bit-exact against the numpy restatement (oracle/np_oracle.py, itself pinned by those fixtures)
    on graphs the fixtures cannot carry;
  * for the random draws: the properties the reference guarantees (counts, distinctness, membership,
    relabelling, row order), reproducibility under torch.manual_seed, and chi-square uniformity.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEV = 'cuda'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope='module')
def ts():
    import pytorch_sparse_amd
    return pytorch_sparse_amd


@pytest.fixture(scope='module')
def graph(ts):
    G = np.load(os.path.join(GOLDEN, 'py4_graph.npz'))
    A = ts.SparseTensor(row=dev(G['row']), col=dev(G['col']), value=dev(G['value']),
                        sparse_sizes=(int(G['n']), int(G['n'])), is_sorted=True)
    return G, A


def _fixtures(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + '*.npz')))


@pytest.mark.parametrize('path', _fixtures('py4_rw_'), ids=os.path.basename)
def test_golden_random_walk(graph, path):
    G, A = graph
    z = np.load(path)
    out = torch.ops.tsamd.random_walk_with_rand(dev(G['rowptr']), dev(G['col']), dev(z['start']), dev(z['rand']))
    np.testing.assert_array_equal(host(out), z['out'])


@pytest.mark.parametrize('path', _fixtures('py4_sample_all_'), ids=os.path.basename)
def test_golden_sample_adj_take_all(ts, graph, path):
    G, A = graph
    z = np.load(path)
    rp, c, n_id, e_id = torch.ops.torch_sparse.sample_adj(dev(G['rowptr']), dev(G['col']), dev(z['idx']), -1, False)
    for got, key in ((rp, 'rowptr'), (c, 'col'), (n_id, 'n_id'), (e_id, 'e_id')):
        np.testing.assert_array_equal(host(got), z[key], err_msg=key)
    adj, n_id2 = A.sample_adj(dev(z['idx']), -1)
    assert torch.equal(n_id2, n_id)
    assert adj.sparse_sizes() == (z['idx'].size, z['n_id'].size)
    np.testing.assert_array_equal(host(adj.storage.value()), z['value'])


@pytest.mark.parametrize('path', _fixtures('py4_saint_'), ids=os.path.basename)
def test_golden_saint_subgraph(ts, graph, path):
    G, A = graph
    z = np.load(path)
    r, c, e = torch.ops.torch_sparse.saint_subgraph(dev(z['idx']), dev(G['rowptr']), dev(G['row']), dev(G['col']))
    for got, key in ((r, 'row'), (c, 'col'), (e, 'edge_index')):
        np.testing.assert_array_equal(host(got), z[key], err_msg=key)
    sub, e2 = ts.saint_subgraph(A, dev(z['idx']))
    assert torch.equal(e2, e) and sub.sparse_sizes() == (z['idx'].size, z['idx'].size)
    np.testing.assert_array_equal(host(sub.storage.value()), z['value'])


@pytest.mark.parametrize('path', _fixtures('py4_relabel_'), ids=os.path.basename)
def test_golden_relabel(graph, path):
    G, A = graph
    z = np.load(path)
    if 'one_hop' in path:
        bip = path.endswith('_1.npz')
        rp, c, v, oi = torch.ops.torch_sparse.relabel_one_hop(dev(G['rowptr']), dev(G['col']), dev(G['value']),
                                                              dev(z['idx']), bip)
        np.testing.assert_array_equal(host(rp), z['rowptr'])
        np.testing.assert_array_equal(host(c), z['col'])
        np.testing.assert_array_equal(host(v), z['value'])
        np.testing.assert_array_equal(host(oi), z['out_idx'])
        rp2, c2, v2, _ = torch.ops.torch_sparse.relabel_one_hop(dev(G['rowptr']), dev(G['col']), None,
                                                                dev(z['idx']), bip)
        assert v2 is None and torch.equal(c2, c) and torch.equal(rp2, rp)
    else:
        oc, oi = torch.ops.torch_sparse.relabel(dev(z['col']), dev(z['idx']))
        np.testing.assert_array_equal(host(oc), z['out_col'])
        np.testing.assert_array_equal(host(oi), z['out_idx'])


# ---------------------------------------------------------------------------------------------
# larger graphs against the numpy restatement
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def big():
    from pytorch_sparse_amd import synth
    rowptr, col = synth.rmat_csr(17, 16, seed=5)
    return rowptr.numpy(), col.numpy()


def test_large_deterministic_ops_match_oracle(big):
    rowptr, col = big
    n = rowptr.size - 1
    rng = np.random.default_rng(0)
    rp, c = dev(rowptr), dev(col)
    row = torch.ops.torch_sparse.ptr2ind(rp, col.size)

    idx = rng.permutation(n)[:20_000]
    got = torch.ops.torch_sparse.sample_adj(rp, c, dev(idx), -1, False)
    want = npo.sample_adj_all(rowptr, col, idx)
    # R-MAT has duplicate edges: equal new column ids inside a row may swap their e_id (the
    # reference's std::sort is not stable either) -> compare e_id through the column it points to
    for g, w, key in zip(got[:3], want[:3], ('rowptr', 'col', 'n_id')):
        np.testing.assert_array_equal(host(g), w, err_msg=key)
    np.testing.assert_array_equal(col[host(got[3])], col[want[3]])
    np.testing.assert_array_equal(np.sort(host(got[3])), np.sort(want[3]))

    sub = rng.permutation(n)[:40_000]
    got = torch.ops.torch_sparse.saint_subgraph(dev(sub), rp, row, c)
    for g, w, key in zip(got, npo.saint_subgraph(sub, rowptr, col), ('row', 'col', 'edge_index')):
        np.testing.assert_array_equal(host(g), w, err_msg=key)

    cols = col[rng.integers(0, col.size, 300_000)]
    got = torch.ops.torch_sparse.relabel(dev(cols), dev(idx))
    for g, w in zip(got, npo.relabel(cols, idx)):
        np.testing.assert_array_equal(host(g), w)

    for bip in (False, True):
        g_rp, g_c, _, g_idx = torch.ops.torch_sparse.relabel_one_hop(rp, c, None, dev(idx), bip)
        w_rp, w_c, _, w_idx = npo.relabel_one_hop(rowptr, col, idx, bip)
        np.testing.assert_array_equal(host(g_rp), w_rp)
        np.testing.assert_array_equal(host(g_c), w_c)
        np.testing.assert_array_equal(host(g_idx), w_idx)

    start = rng.integers(0, n, 100_000)
    rand = rng.random((100_000, 20), dtype=np.float32)
    out = torch.ops.tsamd.random_walk_with_rand(rp, c, dev(start), dev(rand))
    np.testing.assert_array_equal(host(out), npo.random_walk(rowptr, col, start, rand))


def test_random_walk_follows_edges(ts, big):
    rowptr, col = big
    n = rowptr.size - 1
    A = ts.SparseTensor(rowptr=dev(rowptr), col=dev(col), sparse_sizes=(n, n), is_sorted=True)
    start = torch.randint(0, n, (50_000, ), device=DEV)
    torch.manual_seed(3)
    walk = A.random_walk(start, 8)
    assert walk.shape == (50_000, 9) and torch.equal(walk[:, 0], start)
    torch.manual_seed(3)
    assert torch.equal(ts.random_walk(A, start, 8), walk)  # reproducible under the device generator
    w = host(walk)
    deg = rowptr[1:] - rowptr[:-1]
    keys = set((np.repeat(np.arange(n), deg) * n + col).tolist())
    src, dst = w[:, :-1].ravel(), w[:, 1:].ravel()
    stay = deg[src] == 0
    assert np.array_equal(dst[stay], src[stay])  # isolated nodes keep the walk in place
    assert all(k in keys for k in (src[~stay] * n + dst[~stay]).tolist())
    assert len(np.unique(w[:, 1])) > 1000  # not degenerate


@pytest.mark.parametrize('replace', [False, True])
@pytest.mark.parametrize('k', [1, 5, 25])
def test_sample_adj_properties(ts, big, k, replace):
    rowptr, col = big
    n = rowptr.size - 1
    rng = np.random.default_rng(k)
    idx = rng.permutation(n)[:30_000]
    A = ts.SparseTensor(rowptr=dev(rowptr), col=dev(col), value=torch.arange(col.size, device=DEV).float(),
                        sparse_sizes=(n, n), is_sorted=True)
    torch.manual_seed(11)
    adj, n_id = A.sample_adj(dev(idx), k, replace=replace)
    rp, c, v = (host(t) for t in adj.csr())
    n_id = host(n_id)
    e_id = v.astype(np.int64)  # value = position in the source, so it doubles as e_id
    deg = rowptr[idx + 1] - rowptr[idx]
    cnt = np.where(deg > 0, k, 0) if replace else np.minimum(deg, k)
    np.testing.assert_array_equal(rp[1:] - rp[:-1], cnt)
    assert adj.sparse_sizes() == (idx.size, n_id.size)
    # relabelling: seeds first, every id once, columns point back at the sampled source entry
    np.testing.assert_array_equal(n_id[:idx.size], idx)
    assert np.unique(n_id).size == n_id.size
    np.testing.assert_array_equal(n_id[c], col[e_id])
    # every sampled entry belongs to the row of its seed
    seg = np.repeat(np.arange(idx.size), cnt)
    assert np.all((e_id >= rowptr[idx][seg]) & (e_id < rowptr[idx + 1][seg]))
    if not replace:
        assert np.unique(e_id).size == e_id.size  # distinct draws
    # rows sorted by the new column id
    same_row = seg[1:] == seg[:-1]
    assert np.all(c[1:][same_row] >= c[:-1][same_row])
    # new nodes are numbered in first-occurrence order of the (row-major) draws: the first time an id
    # >= n_seeds appears when rows are scanned in order, it must be the next unused id
    # (checked on the unsorted draw order through the op's e_id is not possible; check monotone cover)
    new = c[c >= idx.size]
    assert new.size == 0 or (np.unique(new).size == n_id.size - idx.size)
    # reproducible
    torch.manual_seed(11)
    adj2, n_id2 = A.sample_adj(dev(idx), k, replace=replace)
    assert torch.equal(adj2.storage.col(), adj.storage.col()) and np.array_equal(host(n_id2), n_id)
    torch.manual_seed(12)
    adj3, _ = A.sample_adj(dev(idx), k, replace=replace)
    if (deg > k).sum() > 100:
        assert not torch.equal(adj3.storage.value(), adj.storage.value())


@pytest.mark.parametrize('replace', [False, True])
def test_sample_adj_uniform(ts, replace):
    """2000 rows share the same 1000 neighbours; every neighbour must be drawn equally often."""
    R, D, k = 2000, 1000, 100
    rowptr = np.concatenate([np.arange(R + 1) * D, np.full(D, R * D)]).astype(np.int64)
    col = np.tile(np.arange(R, R + D), R).astype(np.int64)
    torch.manual_seed(2024)
    rp, c, n_id, e_id = torch.ops.torch_sparse.sample_adj(dev(rowptr), dev(col), torch.arange(R, device=DEV), k, replace)
    picked = col[host(e_id)] - R
    counts = np.bincount(picked, minlength=D).astype(np.float64)
    expect = R * k / D
    chi2 = ((counts - expect) ** 2 / expect).sum()
    # with replacement chi2 ~ chi-square(999): 999 +- 45; without, the draws of a row are negatively
    # correlated and the statistic shrinks by (1 - k/D)
    lo, hi = (820, 1180) if replace else (720, 1080)
    assert lo < chi2 < hi, chi2
    # different rows must draw independently: two rows share k*k/D = 10 neighbours on average
    B = np.zeros((R, D), bool)
    B[np.repeat(np.arange(R), k), picked] = True
    overlap = (B[:-1] & B[1:]).sum(1).mean()
    want = k * k / D if not replace else D * (1 - (1 - 1 / D) ** k) ** 2
    assert abs(overlap - want) < 0.5, overlap


def test_sample_with_replacement_api(ts, big):
    rowptr, col = big
    n = rowptr.size - 1
    A = ts.SparseTensor(rowptr=dev(rowptr), col=dev(col), sparse_sizes=(n, n), is_sorted=True)
    subset = torch.randint(0, n, (10_000, ), device=DEV)
    out = host(A.sample(7, subset))
    assert out.shape == (10_000, 7)
    s = host(subset)
    deg = rowptr[s + 1] - rowptr[s]
    assert np.all(out[deg == 0] == -1)
    for i in np.nonzero(deg > 0)[0][:500]:
        assert set(out[i]) <= set(col[rowptr[s[i]]:rowptr[s[i] + 1]])
    assert ts.sample(A, 3).shape == (n, 3)


def test_bad_ids_raise(ts, graph):
    G, A = graph
    n = int(G['n'])
    bad = torch.tensor([0, n], device=DEV)
    with pytest.raises(IndexError):
        A.sample_adj(bad, 3)
    with pytest.raises(IndexError):
        A.saint_subgraph(bad)
    with pytest.raises(IndexError):
        torch.ops.torch_sparse.relabel_one_hop(dev(G['rowptr']), dev(G['col']), None, bad, False)


# ---------------------------------------------------------------------------------------------
# multi-hop neighbor_sample (PyG's NeighborLoader entry point)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('path', _fixtures('py5_neighbor_sample_'), ids=os.path.basename)
def test_golden_neighbor_sample(path):
    z = np.load(path)
    got = torch.ops.torch_sparse.neighbor_sample(dev(z['colptr']), dev(z['row']), dev(z['input_node']),
                                                 z['num_neighbors'].tolist(), False, bool(z['directed']))
    for g, key in zip(got, ('node', 'out_row', 'out_col', 'out_edge')):
        np.testing.assert_array_equal(host(g), z[key], err_msg=key)


def _csc(big):
    rowptr, col = big  # use the CSR arrays of the R-MAT graph as a CSC (colptr, row): same structure
    return rowptr, col


@pytest.mark.parametrize('directed', [True, False])
def test_large_neighbor_sample_take_all_matches_oracle(big, directed):
    colptr, row = _csc(big)
    n = colptr.size - 1
    inp = np.random.default_rng(1).permutation(n)[:200]
    got = torch.ops.torch_sparse.neighbor_sample(dev(colptr), dev(row), dev(inp), [-1, -1], False, directed)
    want = npo.neighbor_sample_all(colptr, row, inp, 2, directed)
    for g, w, key in zip(got, want, ('node', 'row', 'col', 'edge')):
        np.testing.assert_array_equal(host(g), w, err_msg=key)


@pytest.mark.parametrize('replace', [False, True])
def test_neighbor_sample_properties(big, replace):
    colptr, row = _csc(big)
    n = colptr.size - 1
    inp = np.random.default_rng(2).permutation(n)[:1024]
    fan = [10, 5, 3]
    torch.manual_seed(7)
    node, r, c, e = (host(t) for t in torch.ops.torch_sparse.neighbor_sample(dev(colptr), dev(row), dev(inp), fan,
                                                                             replace, True))
    # seeds first, every node once, edges consistent with the source arrays
    np.testing.assert_array_equal(node[:inp.size], inp)
    assert np.unique(node).size == node.size
    np.testing.assert_array_equal(node[r], row[e])                       # source of the sampled entry
    assert np.all((e >= colptr[node[c]]) & (e < colptr[node[c] + 1]))      # entry lies in the target's column
    assert r.max() < node.size and c.max() < node.size
    # hop structure: targets of hop l are exactly the nodes discovered in hop l-1, in order
    deg = colptr[1:] - colptr[:-1]
    begin, end, off = 0, inp.size, 0
    for k in fan:
        d = deg[node[begin:end]]
        cnt = np.where(d > 0, k, 0) if replace else np.minimum(d, k)
        T = int(cnt.sum())
        np.testing.assert_array_equal(c[off:off + T], np.repeat(np.arange(begin, end), cnt))
        if not replace:
            assert np.unique(e[off:off + T]).size == T
        new_end = int(max(r[off:off + T].max(initial=end - 1) + 1, end))
        # first-occurrence numbering: ids >= end appear in increasing order of first appearance
        fresh = r[off:off + T][r[off:off + T] >= end]
        first = fresh[np.sort(np.unique(fresh, return_index=True)[1])]
        np.testing.assert_array_equal(first, np.arange(end, new_end))
        begin, end, off = end, new_end, off + T
    assert off == e.size and end == node.size
    # reproducible, and undirected mode returns the induced sub-graph of the same node set
    torch.manual_seed(7)
    node2, r2, c2, e2 = torch.ops.torch_sparse.neighbor_sample(dev(colptr), dev(row), dev(inp), fan, replace, False)
    np.testing.assert_array_equal(host(node2), node)
    i, v, pos = npo.saint_subgraph(node, colptr, row)
    np.testing.assert_array_equal(host(r2), v)
    np.testing.assert_array_equal(host(c2), i)
    np.testing.assert_array_equal(host(e2), pos)


def test_duplicate_seed_ids_keep_their_last_position(big):
    """The reference's sequential map insert lets a node that is listed twice keep its LAST position;
    the numpy restatement agrees with the compiled reference on that (checked on the CPU), and so
    must the GPU relabel / SAINT kernels."""
    rowptr, col = big
    n = rowptr.size - 1
    rng = np.random.default_rng(9)
    idx = rng.integers(0, n, 5_000)  # duplicates
    assert np.unique(idx).size < idx.size
    rp, c = dev(rowptr), dev(col)
    row = torch.ops.torch_sparse.ptr2ind(rp, col.size)
    cols = col[rng.integers(0, col.size, 100_000)]
    for g, w in zip(torch.ops.torch_sparse.relabel(dev(cols), dev(idx)), npo.relabel(cols, idx)):
        np.testing.assert_array_equal(host(g), w)
    for g, w in zip(torch.ops.torch_sparse.saint_subgraph(dev(idx), rp, row, c), npo.saint_subgraph(idx, rowptr, col)):
        np.testing.assert_array_equal(host(g), w)
    g_rp, g_c, _, g_idx = torch.ops.torch_sparse.relabel_one_hop(rp, c, None, dev(idx), True)
    w_rp, w_c, _, w_idx = npo.relabel_one_hop(rowptr, col, idx, True)
    np.testing.assert_array_equal(host(g_c), w_c)
    np.testing.assert_array_equal(host(g_idx), w_idx)


@pytest.mark.parametrize('D,k', [(3, 1), (3, 2), (5, 2), (8, 3), (8, 5), (16, 4), (64, 2), (65, 2), (100, 2), (129, 2),
                                 (70, 3)])
def test_sample_adj_every_subset_equally_likely(D, k):
    """Stronger than equal inclusion frequencies: all C(D, k) k-subsets of a row must be drawn equally
    often (chi-square over the subsets).  Rows with <= 64 neighbours use Floyd's exact algorithm,
    longer rows the keyed Feistel bijection; a first version of the bijection (multiply / xor-shift
    rounds) passed the inclusion test and failed this one by orders of magnitude."""
    import itertools
    R = 400_000 if D <= 64 else 2_000_000
    rowptr = np.concatenate([np.arange(R + 1) * D, np.full(D, R * D)]).astype(np.int64)
    col = np.tile(np.arange(R, R + D), R).astype(np.int64)
    torch.manual_seed(1)
    _, _, _, e_id = torch.ops.torch_sparse.sample_adj(dev(rowptr), dev(col), torch.arange(R, device=DEV), k, False)
    pos = np.sort((host(e_id) % D).reshape(R, k), axis=1)
    assert np.all(pos[:, 1:] > pos[:, :-1]) if k > 1 else True  # distinct
    key = np.zeros(R, np.int64)
    for j in range(k):
        key = key * D + pos[:, j]
    nsub = len(list(itertools.combinations(range(D), k))) if D <= 16 else int(np.prod([D - j for j in range(k)]) // np.prod(range(1, k + 1)))
    _, counts = np.unique(key, return_counts=True)
    cnt = np.zeros(nsub)
    cnt[:counts.size] = counts
    expect = R / nsub
    chi2 = ((cnt - expect) ** 2 / expect).sum()
    dof = nsub - 1
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof) + 5, (chi2, dof)


def test_relabel_seed_extend_cabi_matches_a_sequential_map():
    """tsamd_relabel_seed / tsamd_relabel_extend (the multi-hop samplers' persistent relabel, include/tsamd.h): three
    successive extends against the sequential std::unordered_map walk of neighbor_sample_cpu.cpp:41-103 restated as a
    Python dict -- ids, list order and the device counter, with no read-back between the calls."""
    import ctypes
    from pytorch_sparse_amd import _native as nat
    L = nat.lib()
    L.tsamd_relabel_workspace_bytes.restype = ctypes.c_size_t
    g = torch.Generator().manual_seed(7)
    M = 5000
    seeds = torch.randperm(M, generator=g)[:300]
    draws = [torch.randint(0, M, (n, ), generator=g) for n in (2000, 1, 7000)]
    # host restatement
    to_local = {int(v): i for i, v in enumerate(seeds.tolist())}
    order = seeds.tolist()
    want_local = []
    for d in draws:
        loc = []
        for v in d.tolist():
            if v not in to_local:
                to_local[v] = len(order)
                order.append(v)
            loc.append(to_local[v])
        want_local.append(loc)
    # device
    cap = seeds.numel() + sum(d.numel() for d in draws)
    buf = torch.empty(cap, dtype=torch.long, device=DEV)
    buf[:seeds.numel()] = seeds.to(DEV)
    slot = torch.empty(M, dtype=torch.long, device=DEV)
    count = torch.empty(1, dtype=torch.long, device=DEV)
    err = torch.empty(1, dtype=torch.long, device=DEV)
    st = nat.stream_ptr(buf.device)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    I = lambda x: ctypes.c_int64(int(x))  # noqa: E731
    nat.check(L.tsamd_relabel_seed(P(buf), I(seeds.numel()), I(M), P(slot), P(count), P(err), ctypes.c_int(1), st), 'seed')
    got_local = []
    for d in draws:
        dd = d.to(DEV)
        T = dd.numel()
        rank = torch.empty(T + 1, dtype=torch.long, device=DEV)
        local = torch.empty(T, dtype=torch.long, device=DEV)
        ws = nat.workspace(L.tsamd_relabel_workspace_bytes(I(T)), buf.device)
        nat.check(L.tsamd_relabel_extend(P(dd), I(T), I(M), P(slot), P(rank), P(count), P(local), P(buf), I(cap), P(err),
                                         P(ws), ctypes.c_size_t(ws.numel()), st), 'extend')
        got_local.append(local)
    assert int(err.item()) == 0
    assert int(count.item()) == len(order)
    assert buf[:len(order)].cpu().tolist() == order
    for gl, wl in zip(got_local, want_local):
        assert gl.cpu().tolist() == wl
    # an id outside [0, M) and an append beyond the capacity are counted, not written
    bad = torch.tensor([M, -1, 3], dtype=torch.long, device=DEV)
    rank = torch.empty(4, dtype=torch.long, device=DEV)
    ws = nat.workspace(L.tsamd_relabel_workspace_bytes(I(3)), buf.device)
    nat.check(L.tsamd_relabel_extend(P(bad), I(3), I(M), P(slot), P(rank), P(count), None, P(buf), I(cap), P(err), P(ws),
                                     ctypes.c_size_t(ws.numel()), st), 'extend')
    assert int(err.item()) == 2

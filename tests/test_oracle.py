"""Pin the oracle: reference golden vectors, committed reference-generated fixtures, and (when
built) the compiled reference itself.  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as oc
from oracle import ref
from pytorch_sparse_amd import synth
from tests.util import ALL_DTYPES, CODE, fromnp, tonp

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_readme_spmm_vector():
    # reference test/test_spmm.py:10-19 == README.md:225-238 (all six dtypes)
    rowptr, col = [0, 2, 3, 5], [0, 2, 1, 0, 1]
    val, x = [1, 2, 4, 1, 3], [[1, 4], [2, 5], [3, 6]]
    for code in (oc.F32, oc.F64, oc.F16, oc.I32, oc.I64):
        out, _ = oc.spmm(code, 'sum', rowptr, col, val, x)
        assert out.tolist() == [[7, 16], [8, 20], [7, 19]]
    out, _ = oc.spmm(oc.BF16, 'sum', rowptr, col, oc.f32_to_bf16_bits(np.float32(val)),
                     oc.f32_to_bf16_bits(np.float32(x)))
    assert oc.bf16_bits_to_f32(out).tolist() == [[7, 16], [8, 20], [7, 19]]


def test_survey_pinned_semantics():
    # SURVEY.md 8c, probed on the compiled reference: ties -> first edge, empty row -> 0 / arg=E
    rp, c, x = [0, 3, 3, 5], [0, 1, 2, 0, 0], [[1, 5], [1, 7], [1, 7]]
    out, arg = oc.spmm(oc.F32, 'max', rp, c, None, x)
    assert out.tolist() == [[1, 7], [0, 0], [1, 5]] and arg.tolist() == [[0, 1], [5, 5], [3, 3]]
    out, arg = oc.spmm(oc.F32, 'min', rp, c, None, x)
    assert arg.tolist() == [[0, 0], [5, 5], [3, 3]]
    out, _ = oc.spmm(oc.F32, 'mean', rp, c, None, x)
    assert np.allclose(out, [[1, 6.3333335], [0, 0], [1, 5]], rtol=0, atol=1e-7)
    # max with the reference's example values: out [[6,12],[8,20],[6,15]], arg [[1,1],[2,2],[4,4]]
    out, arg = oc.spmm(oc.F32, 'max', [0, 2, 3, 5], [0, 2, 1, 0, 1], [1, 2, 4, 1, 3],
                       [[1, 4], [2, 5], [3, 6]])
    assert out.tolist() == [[6, 12], [8, 20], [6, 15]] and arg.tolist() == [[1, 1], [2, 2], [4, 4]]


def test_nan_semantics():
    # NaN in X: never wins a max/min (strict compare), propagates through sum
    rp, c = [0, 3], [0, 1, 2]
    x = np.array([[1.0], [np.nan], [3.0]], dtype=np.float32)
    out, arg = oc.spmm(oc.F32, 'max', rp, c, None, x)
    assert out.tolist() == [[3.0]] and arg.tolist() == [[2]]
    out, _ = oc.spmm(oc.F32, 'sum', rp, c, None, x)
    assert np.isnan(out).all()


def test_narrow_accumulation_is_sequential():
    # fp16 sum of 4096 U(0,1) accumulates in fp16 (!= exact); bf16 saturates at 256
    rp, c = [0, 4096], np.zeros(4096, dtype=np.int64)
    ones = np.ones((1, 1), dtype=np.float32)
    out, _ = oc.spmm(oc.BF16, 'sum', rp, c, None, oc.f32_to_bf16_bits(ones))
    assert oc.bf16_bits_to_f32(out).item() == 256.0
    out, _ = oc.spmm(oc.BF16, 'sum', rp, c, None, oc.f32_to_bf16_bits(ones), wide_acc=True)
    assert oc.bf16_bits_to_f32(out).item() == 4096.0


def test_ind2ptr_ptr2ind_vectors():
    # reference test/test_storage.py:10-24
    assert oc.ind2ptr([2, 2, 4, 5, 5, 6], 8).tolist() == [0, 0, 0, 2, 2, 3, 5, 6, 6]
    assert oc.ptr2ind([0, 0, 0, 2, 2, 3, 5, 6, 6], 6).tolist() == [2, 2, 4, 5, 5, 6]
    assert oc.ind2ptr([], 4).tolist() == [0, 0, 0, 0, 0]
    assert oc.ptr2ind([0, 0, 0], 0).tolist() == []


@pytest.mark.skipif(not ref.available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('dtype', ALL_DTYPES)
@pytest.mark.parametrize('reduce', ['sum', 'mean', 'min', 'max'])
def test_c_oracle_matches_compiled_reference(dtype, reduce):
    r = ref.ops()
    rp, c = synth.rmat_csr(9, 8, seed=3)
    E, n = c.numel(), 512
    if dtype.is_floating_point:
        v = synth.values(E, dtype=dtype)
        x = synth.features(n, 7, dtype=dtype, batch=(2, ))
    else:
        g = torch.Generator().manual_seed(0)
        lo = 0 if dtype == torch.uint8 else 1  # unsigned: no negative draws
        v = torch.randint(-5 * lo, 5, (E, ), dtype=dtype, generator=g)
        x = torch.randint(-9 * lo, 9, (2, n, 7), dtype=dtype, generator=g)
        # 8- / 16-bit sums wrap, and the reference's mean divides by the count cast to the type:
        # a row whose length is a multiple of 256 would be a division by zero (SIGFPE) there
        deg = rp[1:] - rp[:-1]
        assert not bool(((deg % 256 == 0) & (deg > 0)).any())
    for value in (v, None):
        if reduce == 'sum':
            ro, ra = r.spmm_sum(None, rp, c, value, None, None, x), None
        elif reduce == 'mean':
            ro, ra = r.spmm_mean(None, rp, c, value, None, None, None, x), None
        elif reduce == 'min':
            ro, ra = r.spmm_min(rp, c, value, x)
        else:
            ro, ra = r.spmm_max(rp, c, value, x)
        co, ca = oc.spmm(CODE[dtype], reduce, rp.numpy(), c.numpy(), tonp(value), tonp(x))
        assert np.array_equal(tonp(ro), co)  # bit-exact, narrow types included
        if ra is not None:
            # where no entry beat the reducer's init value (all candidates equal numeric_limits::max /
            # lowest: e.g. an all-zero uint8 row under max) the reference leaves a stale index behind --
            # unspecified; the oracle (and the HIP path) report E there
            deg = (rp[1:] - rp[:-1]).numpy()
            unspecified = (ca == E) & (deg[None, :, None] > 0)
            assert np.array_equal(np.where(unspecified, E, ra.numpy()), ca)


@pytest.mark.skipif(not ref.available(), reason='oracle/_ref not built')
def test_c_oracle_convert_matches_compiled_reference():
    r = ref.ops()
    g = torch.Generator().manual_seed(1)
    ind, _ = torch.sort(torch.randint(0, 300, (2000, ), generator=g))
    ptr = r.ind2ptr(ind, 300)
    assert np.array_equal(ptr.numpy(), oc.ind2ptr(ind.numpy(), 300))
    assert np.array_equal(r.ptr2ind(ptr, 2000).numpy(), oc.ptr2ind(ptr.numpy(), 2000))


@pytest.mark.skipif(not ref.available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('reduce', ['sum', 'mean'])
def test_c_oracle_value_bw_matches_compiled_reference(reduce):
    # the reference reaches spmm_value_bw only through autograd (csrc/spmm.cpp:96-98)
    r = ref.ops()
    rp, c = synth.rmat_csr(8, 6, seed=4)
    E, n = c.numel(), 256
    row = torch.from_numpy(oc.ptr2ind(rp.numpy(), E))
    for dtype in (torch.float32, torch.float64):
        v = synth.values(E, dtype=dtype).requires_grad_()
        x = synth.features(n, 6, dtype=dtype, batch=(2, ))
        gout = synth.features(n, 6, seed=3, dtype=dtype, batch=(2, ))
        if reduce == 'sum':
            out = r.spmm_sum(row, rp, c, v, None, None, x)
        else:
            out = r.spmm_mean(row, rp, c, v, None, None, None, x)
        out.backward(gout)
        got = oc.spmm_value_bw(CODE[dtype], reduce, row.numpy(), rp.numpy(), c.numpy(), tonp(x),
                               tonp(gout))
        assert np.array_equal(tonp(v.grad), got)


def test_golden_fixtures():
    """Fixtures written by tests/golden/make_golden.py from the reference's own code."""
    files = sorted(glob.glob(os.path.join(GOLDEN, 'spmm_*.npz')))
    assert files, 'no golden fixtures committed'
    for f in files:
        z = np.load(f)
        code, reduce = int(z['dtype_code']), str(z['reduce'])
        value = z['value'] if 'value' in z.files else None
        out, arg = oc.spmm(code, reduce, z['rowptr'], z['col'], value, z['mat'])
        assert np.array_equal(out, z['out']), f
        if arg is not None:
            assert np.array_equal(arg, z['arg_out']), f


def test_np_oracle_matches_reference_python_fixtures():
    """coalesce / transpose / storage / spspmm restatements vs outputs of the reference's own
    Python package (tests/golden/py_*.npz, written by make_golden.py)."""
    from oracle import np_oracle as no
    n_checked = 0
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_coalesce_*.npz'))):
        z = np.load(f)
        value = z['value'] if 'value' in z.files else None
        op = str(z['op']) if 'op' in z.files else 'add'
        r, c, v = no.coalesce(z['index'][0], z['index'][1], value, int(z['m']), int(z['n']), op)
        assert np.array_equal(np.stack([r, c]), z['out_index']), f
        if value is not None:
            assert np.array_equal(v, z['out_value']), f
        n_checked += 1
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_transpose_*.npz'))):
        z = np.load(f)
        r, c, v = no.transpose(z['index'][0], z['index'][1], z['value'], int(z['m']), int(z['n']))
        assert np.array_equal(np.stack([r, c]), z['out_index']) and np.array_equal(v, z['out_value']), f
        n_checked += 1
    z = np.load(os.path.join(GOLDEN, 'py_storage.npz'))
    m, n = int(z['m']), int(z['n'])
    r, c, perm = no.sort_coo(z['row'], z['col'], m, n)
    assert np.array_equal(r, z['s_row']) and np.array_equal(c, z['s_col'])
    assert np.array_equal(z['value'][perm], z['s_value'])
    assert np.array_equal(oc.ind2ptr(r, m), z['rowptr'])
    p = no.csr2csc(r, c, m, n)
    assert np.array_equal(p, z['csr2csc'])
    assert np.array_equal(np.argsort(p, kind='stable'), z['csc2csr'])
    assert np.array_equal(oc.ind2ptr(c[p], n), z['colptr'])
    assert np.array_equal(c[p], z['t_row']) and np.array_equal(r[p], z['t_col'])
    for f in sorted(glob.glob(os.path.join(GOLDEN, 'py_spspmm_*.npz'))):
        z = np.load(f)
        vA = z['vA'] if 'vA' in z.files else None
        vB = z['vB'] if 'vB' in z.files else None
        r, c, v = no.spspmm(z['iA'][0], z['iA'][1], vA, z['iB'][0], z['iB'][1], vB, int(z['m']),
                            int(z['k']), int(z['n']))
        assert np.array_equal(np.stack([r, c]), z['iC']), f
        if vA is not None:
            assert np.allclose(v, z['vC'], rtol=1e-6, atol=1e-6), f
        n_checked += 1
    assert n_checked > 20


def test_np_oracle_matches_reference_sampler_fixtures():
    """random_walk / relabel / relabel_one_hop / take-all sample_adj / saint_subgraph restatements
    against the outputs of the reference's CPU kernels compiled unmodified (make_golden.py part 4)."""
    import glob
    from oracle import np_oracle as npo
    G = np.load(os.path.join(GOLDEN, 'py4_graph.npz'))
    rowptr, col = G['rowptr'], G['col']
    n_checked = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, 'py4_*.npz'))):
        name = os.path.basename(path)
        z = np.load(path)
        if name.startswith('py4_rw_'):
            np.testing.assert_array_equal(npo.random_walk(rowptr, col, z['start'], z['rand']), z['out'])
        elif name.startswith('py4_sample_all_'):
            rp, c, n_id, e_id = npo.sample_adj_all(rowptr, col, z['idx'])
            for got, key in ((rp, 'rowptr'), (c, 'col'), (n_id, 'n_id'), (e_id, 'e_id')):
                np.testing.assert_array_equal(got, z[key], err_msg=name + ':' + key)
        elif name.startswith('py4_saint_'):
            r, c, e = npo.saint_subgraph(z['idx'], rowptr, col)
            for got, key in ((r, 'row'), (c, 'col'), (e, 'edge_index')):
                np.testing.assert_array_equal(got, z[key], err_msg=name + ':' + key)
        elif name.startswith('py4_relabel_one_hop_'):
            bip = name.endswith('_1.npz')
            rp, c, pos, oi = npo.relabel_one_hop(rowptr, col, z['idx'], bip)
            np.testing.assert_array_equal(rp, z['rowptr'])
            np.testing.assert_array_equal(c, z['col'])
            np.testing.assert_array_equal(oi, z['out_idx'])
            np.testing.assert_array_equal(G['value'][pos], z['value'])
        elif name.startswith('py4_relabel_'):
            oc, oi = npo.relabel(z['col'], z['idx'])
            np.testing.assert_array_equal(oc, z['out_col'])
            np.testing.assert_array_equal(oi, z['out_idx'])
        else:
            continue
        n_checked += 1
    assert n_checked == 24


@pytest.mark.skipif(not ref.available(), reason='oracle/_ref not built')
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_np_oracle_matches_compiled_reference_samplers(seed):
    """The same restatements against the reference's CPU kernels run live (oracle/_ref), on random
    graphs with isolated nodes excluded from walks (the reference reads out of the row there)."""
    from oracle import np_oracle as npo
    r = ref.ops()
    rng = np.random.default_rng(seed)
    n = 300 + 50 * seed
    key = np.unique(np.concatenate([rng.integers(0, n * n, 4000), np.arange(n) * n + (np.arange(n) + 1) % n]))
    row, col = key // n, key % n
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=rowptr[1:])
    t = torch.from_numpy
    idx = rng.permutation(n)[:rng.integers(1, n)]

    torch.manual_seed(seed)
    start = rng.integers(0, n, 77)
    out = r.random_walk(t(rowptr), t(col), t(start), 12)
    torch.manual_seed(seed)
    rand = torch.rand((77, 12)).numpy()
    np.testing.assert_array_equal(npo.random_walk(rowptr, col, start, rand), out.numpy())

    want = r.sample_adj(t(rowptr), t(col), t(idx), -1, False)
    for g, w in zip(npo.sample_adj_all(rowptr, col, idx), want):
        np.testing.assert_array_equal(g, w.numpy())
    want = r.saint_subgraph(t(idx), t(rowptr), t(row), t(col))
    for g, w in zip(npo.saint_subgraph(idx, rowptr, col), want):
        np.testing.assert_array_equal(g, w.numpy())
    cols = col[rng.integers(0, col.size, 500)]
    want = r.relabel(t(cols), t(idx))
    for g, w in zip(npo.relabel(cols, idx), want):
        np.testing.assert_array_equal(g, w.numpy())
    for bip in (False, True):
        w_rp, w_c, _, w_idx = r.relabel_one_hop(t(rowptr), t(col), None, t(idx), bip)
        g_rp, g_c, _, g_idx = npo.relabel_one_hop(rowptr, col, idx, bip)
        np.testing.assert_array_equal(g_rp, w_rp.numpy())
        np.testing.assert_array_equal(g_c, w_c.numpy())
        np.testing.assert_array_equal(g_idx, w_idx.numpy())


def test_np_neighbor_sample_matches_fixtures_and_compiled_reference():
    """Multi-hop take-all neighbour sampling: numpy restatement vs make_golden.py part 5 fixtures, and
    vs the live compiled reference on random graphs."""
    import glob
    from oracle import np_oracle as npo
    paths = sorted(glob.glob(os.path.join(GOLDEN, 'py5_neighbor_sample_*.npz')))
    assert len(paths) == 24
    for path in paths:
        z = np.load(path)
        got = npo.neighbor_sample_all(z['colptr'], z['row'], z['input_node'], len(z['num_neighbors']),
                                      bool(z['directed']))
        for g, key in zip(got, ('node', 'out_row', 'out_col', 'out_edge')):
            np.testing.assert_array_equal(g, z[key], err_msg=os.path.basename(path) + ':' + key)
    if not ref.available():
        return
    r = ref.ops()
    rng = np.random.default_rng(3)
    for trial in range(4):
        n = 500
        key = np.unique(rng.integers(0, n * n, 4000))
        order = np.lexsort((key // n, key % n))
        row, col = (key // n)[order], (key % n)[order]
        colptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(col, minlength=n), out=colptr[1:])
        inp = rng.permutation(n)[:rng.integers(1, 50)]
        for hops in (1, 3):
            for directed in (True, False):
                want = r.neighbor_sample(torch.from_numpy(colptr), torch.from_numpy(row), torch.from_numpy(inp),
                                         [-1] * hops, False, directed)
                for g, w in zip(npo.neighbor_sample_all(colptr, row, inp, hops, directed), want):
                    np.testing.assert_array_equal(g, w.numpy())


HET_NODE_TYPES = ['paper', 'author', 'venue']
HET_EDGE_TYPES = [('author', 'writes', 'paper'), ('paper', 'cites', 'paper'), ('paper', 'in', 'venue'),
                  ('venue', 'hosts', 'paper'), ('paper', 'by', 'author')]
HET_RELS = ['__'.join(e) for e in HET_EDGE_TYPES]


def _random_hetero(rng, sizes, max_deg):
    colptr, row = {}, {}
    for (s, r, d) in HET_EDGE_TYPES:
        deg = rng.integers(0, max_deg + 1, sizes[d])
        deg[::4] = 0
        cp = np.zeros(sizes[d] + 1, np.int64)
        np.cumsum(deg, out=cp[1:])
        colptr['__'.join((s, r, d))] = cp
        row['__'.join((s, r, d))] = rng.integers(0, sizes[s], int(cp[-1]))
    times = {t: rng.integers(0, 50, sizes[t]) for t in HET_NODE_TYPES}
    return colptr, row, times


def test_np_hetero_sampler_matches_fixtures_and_compiled_reference():
    """The sequential restatement of hetero_sample (csrc/cpu/neighbor_sample_cpu.cpp:135-430) against the 88 fixtures
    the compiled reference wrote (make_golden.py part 8) and against the live compiled reference on random graphs --
    with input nodes listed TWICE (the map insert keeps the first position), relations without entries, a node type
    without a time tensor."""
    from oracle import np_oracle as npo
    paths = sorted(glob.glob(os.path.join(GOLDEN, 'py8_hetero_*.npz')))
    assert len(paths) == 88
    for path in paths:
        z = np.load(path)
        colptr = {r: z['colptr__' + r] for r in HET_RELS}
        row = {r: z['row__' + r] for r in HET_RELS}
        inp = {t: z['input__' + t] for t in HET_NODE_TYPES if 'input__' + t in z.files}
        times = {t: z['time__' + t] for t in HET_NODE_TYPES if 'time__' + t in z.files}
        mode, hops, fanv = str(z['mode']), int(z['hops']), int(z['fan'])
        fan = {r: [fanv] * hops for r in HET_RELS}
        node, orow, ocol, oedge = npo.hetero_neighbor_sample_det(
            HET_NODE_TYPES, HET_EDGE_TYPES, colptr, row, inp, fan, hops, mode != 'undirected',
            times if mode.startswith('temporal') else None)
        name = os.path.basename(path)
        for t in HET_NODE_TYPES:
            np.testing.assert_array_equal(node[t], z['node__' + t], err_msg=name + ':node ' + t)
        for r in HET_RELS:
            for got, key in ((orow, 'orow__'), (ocol, 'ocol__'), (oedge, 'oedge__')):
                np.testing.assert_array_equal(got[r], z[key + r], err_msg=name + ':' + key + r)
    if not ref.available() or not hasattr(ref.ops(), 'hetero_neighbor_sample'):
        return
    r_ops = ref.ops()
    rng = np.random.default_rng(11)
    T = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}  # noqa: E731
    for trial in range(6):
        sizes = {'paper': 300, 'author': 120, 'venue': 7}
        colptr, row, times = _random_hetero(rng, sizes, 6)
        if trial % 2:
            row['paper__in__venue'] = row['paper__in__venue'][:0]
            colptr['paper__in__venue'] = np.zeros_like(colptr['paper__in__venue'])
        seeds = rng.integers(0, sizes['paper'], 25)
        seeds[5], seeds[9] = seeds[2], seeds[2]  # listed three times
        inp = {'paper': seeds, 'venue': np.array([3, 3, 1])}
        for hops in (1, 3):
            fan = {r: [-1 if trial % 3 else 50] * hops for r in HET_RELS}
            for directed in (True, False):
                want = r_ops.hetero_neighbor_sample(HET_NODE_TYPES, HET_EDGE_TYPES, T(colptr), T(row), T(inp), fan, hops, False, directed)
                got = npo.hetero_neighbor_sample_det(HET_NODE_TYPES, HET_EDGE_TYPES, colptr, row, inp, fan, hops, directed)
                for t in HET_NODE_TYPES:
                    np.testing.assert_array_equal(got[0][t], want[0][t].numpy())
                for rel in HET_RELS:
                    for k in (1, 2, 3):
                        np.testing.assert_array_equal(got[k][rel], want[k][rel].numpy(), err_msg='%s %d %s' % (rel, k, directed))
            tm = {t: v for t, v in times.items() if t != 'author'} if trial % 2 else times
            want = r_ops.hetero_temporal_neighbor_sample(HET_NODE_TYPES, HET_EDGE_TYPES, T(colptr), T(row), T(inp), fan, T(tm), hops, False, True)
            got = npo.hetero_neighbor_sample_det(HET_NODE_TYPES, HET_EDGE_TYPES, colptr, row, inp, fan, hops, True, tm)
            for t in HET_NODE_TYPES:
                np.testing.assert_array_equal(got[0][t], want[0][t].numpy())
            for rel in HET_RELS:
                for k in (1, 2, 3):
                    np.testing.assert_array_equal(got[k][rel], want[k][rel].numpy())

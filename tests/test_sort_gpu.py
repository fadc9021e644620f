"""The one-sweep radix sort of COO entries (csrc/sort.hip) and the fused duplicate compaction (csrc/coalesce.hip)
against the numpy restatement of the reference's sort-on-construct / coalesce (oracle/np_oracle.py): packed words
(position in the low bits) and key + payload pairs, uniform and power-law inputs (skewed digits), the
device-decided variants on sorted and unsorted input, tile-boundary sizes."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as no
from pytorch_sparse_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    import pytorch_sparse_amd  # noqa: F401
    return torch.ops.tsamd


def _check(ops, row, col, m, n, dev):
    rs, cs, perm = ops.sort_coo(row.to(dev), col.to(dev), m, n, True)
    er, ec, ep = no.sort_coo(row.numpy(), col.numpy(), m, n)
    assert np.array_equal(perm.cpu().numpy(), ep)
    assert np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)
    p2 = ops.sort_coo(row.to(dev), col.to(dev), m, n, False)[2]  # permutation only (csr2csc)
    assert np.array_equal(p2.cpu().numpy(), ep)
    return er, ec, ep


@pytest.mark.parametrize('E,m,n', [
    (8193, 300, 200),                    # just above the one-launch path: 3 tiles, 16-bit keys (packed)
    (4096 * 5, 1 << 10, 1 << 10),        # whole tiles only
    (4096 * 5 + 1, 1 << 10, 1 << 10),    # one entry in the last tile
    (1000003, 500000, 500000),           # 38-bit keys + 20-bit positions: packed, 5 passes
    (3000000, (1 << 21) + 5, (1 << 21) - 3),   # 22 + 21 bits + 22 > 64? no: 65 -> pairs, 6 passes
    (200000, 1 << 31, 1 << 30),          # 61-bit keys: pairs, 8 passes
    (100000, 1, 1),                      # zero key bits: identity
    (50000, 1, 70000),                   # row bits = 0
])
def test_onesweep_sort_bit_exact(dev, ops, E, m, n):
    g = torch.Generator().manual_seed(E % 1013)
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    q = E // 4
    row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()  # duplicates: stability is visible in perm
    _check(ops, row, col, m, n, dev)


def test_onesweep_sort_power_law_and_hot_digits(dev, ops):
    """R-MAT edges in generation order (unsorted; the high row digits take a handful of values -- the wave-uniform
    histogram path) and a degenerate input with ONE distinct key among random ones."""
    row, col = synth.rmat_edges(20, 4, seed=3)
    _check(ops, row, col, 1 << 20, 1 << 20, dev)
    E = 700001
    g = torch.Generator().manual_seed(1)
    row = torch.full((E, ), 77)
    col = torch.full((E, ), 5)
    hot = torch.randint(0, E, (1000, ), generator=g)
    row[hot] = torch.randint(0, 100, (1000, ), generator=g)
    _check(ops, row, col, 100, 100, dev)


@pytest.mark.parametrize('E,m,n', [(75000000, 1 << 22, 1 << 22),   # 44-bit keys + 27-bit positions: pairs, 6 passes
                                   (7500000, 500000, 500000)])        # configs[3]'s input: packed, 5 passes
def test_onesweep_sort_full_size_properties(dev, ops, E, m, n):
    """The BASELINE-size sorts through size-independent properties, all evaluated on the device: keys non-decreasing,
    equal keys in input order (stable), `perm` a permutation, outputs = inputs gathered through it -- and the values that
    ride along (4-byte: through every pass of a packed sort; otherwise gathered by the last pass) equal value[perm]."""
    g = torch.Generator(device=dev).manual_seed(5)
    row = torch.randint(0, m, (E, ), generator=g, device=dev)
    col = torch.randint(0, n, (E, ), generator=g, device=dev)
    val = torch.rand(E, generator=g, device=dev)
    rs, cs, perm, _, vs = ops.sort_coo_values(row, col, m, n, 0, None, val)
    key = rs * n + cs
    d = key[1:] - key[:-1]
    assert bool((d >= 0).all())
    assert bool((perm[1:][d == 0] > perm[:-1][d == 0]).all()), 'equal keys out of input order'
    seen = torch.zeros(E, dtype=torch.bool, device=dev)
    seen[perm] = True
    assert bool(seen.all()), 'perm is not a permutation'
    assert torch.equal(rs, row[perm]) and torch.equal(cs, col[perm])
    assert torch.equal(vs, val[perm])
    v8 = val.double()
    assert torch.equal(ops.sort_coo_values(row, col, m, n, 0, None, v8)[4], v8[perm])  # 8-byte values: last-pass gather


@pytest.mark.parametrize('E,m,n,kind', [
    (131072, 9000, 7000, 'uniform'),            # the smallest input on the bucket path (27-bit keys + 17)
    (400003, 500000, 500000, 'uniform'),        # 38-bit keys, ragged last tile / last bucket
    (1 << 20, 16, 16, 'uniform'),               # 8-bit keys: the bucket id reaches into the position bits
    (900000, 3, 5, 'uniform'),                  # 5-bit keys: three populated rows
    (600000, 1, 1 << 20, 'uniform'),            # no row bits
    (700001, (1 << 19) + 1, 1000, 'uniform'),   # M just above a power of two: half of the buckets stay empty
    (500000, 1 << 20, 1 << 20, 'hub'),          # one row holds a third of the entries: a bucket overflows -> one-sweep passes
    (300000, 1000, 1000, 'onekey'),             # one distinct key among random ones (heavy duplication)
])
def test_bucket_path_bit_exact(dev, ops, E, m, n, kind):
    """Round 6: most-significant-digit scatter + one LDS sort per bucket (and the device-side decision to leave
    it): exact stable permutation, the riding 4-byte values, the 8-byte gather, the probing variants."""
    g = torch.Generator().manual_seed(E % 997)
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    if kind == 'hub':
        row[torch.randperm(E, generator=g)[:E // 3]] = 12345
    elif kind == 'onekey':
        keep = torch.randperm(E, generator=g)[:2000]
        r2, c2 = torch.full((E, ), 77), torch.full((E, ), 5)
        r2[keep], c2[keep] = row[keep], col[keep]
        row, col = r2, c2
    q = E // 5
    row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()  # duplicates: stability is visible in perm
    er, ec, ep = _check(ops, row, col, m, n, dev)
    val = torch.rand(E, generator=g)
    rd, cd, vd = row.to(dev), col.to(dev), val.to(dev)
    key = row.numpy().astype(np.int64) * n + col.numpy()
    want_counts = [int((key[1:] < key[:-1]).sum()), int((key[1:] == key[:-1]).sum())]
    for mode in (0, 1, 3):
        rs, cs, perm, counts, vs = ops.sort_coo_values(rd, cd, m, n, mode, None, vd)
        assert np.array_equal(perm.cpu().numpy(), ep), mode
        assert np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)
        assert np.array_equal(vs.cpu().numpy(), val.numpy()[ep])
        if mode == 1:
            assert counts.tolist() == want_counts
        if mode == 3:
            assert counts.tolist() == want_counts + [int(row.max()), int(col.max())]
    v8 = val.double().to(dev)
    assert np.array_equal(ops.sort_coo_values(rd, cd, m, n, 0, None, v8)[4].cpu().numpy(), val.double().numpy()[ep])
    # already sorted input through the device-decided variant: copy + identity
    rs, cs, perm, counts = ops.sort_coo_auto(torch.from_numpy(er).to(dev), torch.from_numpy(ec).to(dev), m, n)
    assert counts[0].item() == 0 and torch.equal(perm.cpu(), torch.arange(E))
    assert np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)


@pytest.mark.parametrize('E,m,n,kind', [
    (3000000, (1 << 21) + 5, (1 << 21) - 3, 'uniform'),   # 65 bits: keys stripped of the bucket bits, one scatter level
    (12500000, 1 << 21, 1 << 21, 'uniform'),              # 4096 buckets: two scatter levels (6 + 6 bits)
    (24000000, 1 << 22, 1 << 22, 'tiny_l1'),              # 8192 buckets (7 + 6); 40 nearly empty level-1 buckets in front:
])                                                        # the first tile's entries fall outside its counter window
def test_bucket_path_two_levels_bit_exact(dev, ops, E, m, n, kind):
    g = torch.Generator().manual_seed(E % 991)
    if kind == 'tiny_l1':
        few = 6000
        cut = (m * 40) // 128
        row = torch.cat([torch.randint(0, cut, (few, ), generator=g), torch.randint(cut, m, (E - few, ), generator=g)])
        row = row[torch.randperm(E, generator=g)]
    else:
        row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    q = E // 7
    row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()  # duplicates: stability is visible in perm
    er, ec, ep = _check(ops, row, col, m, n, dev)
    val = torch.rand(E, generator=g)
    rs, cs, perm, counts, vs = ops.sort_coo_values(row.to(dev), col.to(dev), m, n, 3, None, val.to(dev))
    assert np.array_equal(perm.cpu().numpy(), ep) and np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)
    assert np.array_equal(vs.cpu().numpy(), val.numpy()[ep])
    assert counts.tolist()[2:] == [int(row.max()), int(col.max())]


@pytest.mark.parametrize('E,m,n,kind', [
    (5000, 300, 200, 'uniform'),                 # the one-launch sort + the compaction kernel
    (400003, 500000, 500000, 'uniform'),         # bucket path: compaction in the bucket sort's output
    (400003, 700, 900, 'uniform'),               # ... with many duplicates per key (runs across the finish step's groups)
    (1 << 20, 16, 16, 'uniform'),                # the bucket id would reach into the position bits: one-sweep + compaction kernel
    (500000, 1 << 20, 1 << 20, 'hub'),           # a bucket overflows: one-sweep + compaction kernel
    (3000000, (1 << 21) + 5, (1 << 21) - 3, 'uniform'),   # stripped keys
    (12500000, 1 << 21, 1 << 21, 'uniform'),     # two scatter levels, 4096 buckets in the look-back
    (300000, 1000, 1000, 'sorted'),              # already in order (with duplicates): copy + compaction
])
@pytest.mark.parametrize('with_value', [False, True])
def test_sort_coalesce_bit_exact(dev, ops, E, m, n, kind, with_value):
    """tsamd::sort_coalesce (sort + duplicate compaction in one op): distinct pairs, run starts, counts and the sorted
    values against the numpy restatement of torch_sparse/storage.py:149-162, 431-447."""
    g = torch.Generator().manual_seed(E % 983 + 1)
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    if kind == 'hub':
        row[torch.randperm(E, generator=g)[:E // 3]] = 12345
    q = E // 5
    row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()
    if kind == 'sorted':
        r_, c_, _ = no.sort_coo(row.numpy(), col.numpy(), m, n)
        row, col = torch.from_numpy(r_.copy()), torch.from_numpy(c_.copy())
    er, ec, ep = no.sort_coo(row.numpy(), col.numpy(), m, n)
    key = er.astype(np.int64) * n + ec
    head = np.ones(E, dtype=bool)
    head[1:] = key[1:] != key[:-1]
    pos = np.nonzero(head)[0]
    ikey = row.numpy().astype(np.int64) * n + col.numpy()
    val = torch.rand(E, generator=g)
    index_u, seg, counts, vs = ops.sort_coalesce(row.to(dev), col.to(dev), m, n, val.to(dev) if with_value else None)
    k = pos.size
    assert counts.tolist() == [int((ikey[1:] < ikey[:-1]).sum()), int((ikey[1:] == ikey[:-1]).sum()), k]
    assert np.array_equal(index_u[0, :k].cpu().numpy(), er[pos]) and np.array_equal(index_u[1, :k].cpu().numpy(), ec[pos])
    assert np.array_equal(seg[:k + 1].cpu().numpy(), np.append(pos, E))
    if with_value:
        assert np.array_equal(vs.cpu().numpy(), val.numpy()[ep])
        v8 = val.double()
        vs8 = ops.sort_coalesce(row.to(dev), col.to(dev), m, n, v8.to(dev))[3]
        assert np.array_equal(vs8.cpu().numpy(), v8.numpy()[ep])


@pytest.mark.parametrize('E,m,n,kind,fused_expected', [
    (5000, 300, 200, 'uniform', False),          # the one-launch sort
    (400003, 500000, 500000, 'uniform', True),   # bucket path, runs of 1-2
    (400003, 700, 900, 'uniform', True),         # bucket path, runs of a few entries (runs across the finish step's groups)
    (400003, 150, 100, 'uniform', True),         # ... of ~27 entries: the bucket sorts by full LSD passes first
    (1 << 20, 16, 16, 'uniform', True),          # one bucket per key: runs of 4096 entries
    (1 << 20, 8, 16, 'uniform', False),          # the bucket id would reach into the position bits: one-sweep + compaction kernel
    (500000, 1 << 20, 1 << 20, 'hub', False),    # a bucket overflows
    (3000000, (1 << 21) + 5, (1 << 21) - 3, 'uniform', True),   # stripped keys
    (300000, 1000, 1000, 'sorted', False),       # already in order
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.int32])
def test_sort_coalesce_reduce_bit_exact(dev, ops, E, m, n, kind, fused_expected, dtype):
    """tsamd::sort_coalesce_reduce (the duplicates' values reduced inside the bucket sort): the same distinct pairs and
    counts as tsamd::sort_coalesce, and -- when the device reports the fused route -- reduced values that equal, bit
    for bit, tsamd::segment_reduce over the sorted values of the unfused op (torch_sparse/storage.py:431-466); both
    routes against numpy's sequential reduceat."""
    from pytorch_sparse_amd.segment import segment_reduce
    g = torch.Generator().manual_seed(E % 977 + 3)
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    if kind == 'hub':
        row[torch.randperm(E, generator=g)[:E // 3]] = 12345
    q = E // 5
    row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()
    if kind == 'sorted':
        r_, c_, _ = no.sort_coo(row.numpy(), col.numpy(), m, n)
        row, col = torch.from_numpy(r_.copy()), torch.from_numpy(c_.copy())
    if dtype == torch.float32:
        val = torch.randn(E, generator=g)
    else:
        val = torch.randint(-1000, 1000, (E, ), generator=g, dtype=torch.int32)
    rd, cd, vd = row.to(dev), col.to(dev), val.to(dev)
    index_0, seg_0, counts_0, vs_0 = ops.sort_coalesce(rd, cd, m, n, vd)
    k = int(counts_0[2])
    saw_fused = False
    for code, op in enumerate(('sum', 'mean', 'min', 'max')):
        index_u, seg, counts, vs, vu = ops.sort_coalesce_reduce(rd, cd, m, n, vd, code)
        c = counts.tolist()
        assert c[:3] == counts_0.tolist() and c[3] in (0, 1)
        assert torch.equal(index_u[:, :k], index_0[:, :k])
        want = segment_reduce(vs_0, None, seg_0, k, op)
        if c[3] == 1:
            saw_fused = True
            got = vu[:k]
        else:
            assert torch.equal(seg[:k + 1], seg_0[:k + 1]) and torch.equal(vs, vs_0)
            got = segment_reduce(vs, None, seg, k, op)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (op, c)
        _, _, ev = no.coalesce(row.numpy(), col.numpy(), val.numpy(), m, n, 'add' if op == 'sum' else op)
        if dtype == torch.int32 or op in ('min', 'max'):
            assert np.array_equal(got.cpu().numpy(), ev)
        else:  # fp32 sums in another association than numpy's: 1e-5 of the run's L1 mass
            _, _, l1 = no.coalesce(row.numpy(), col.numpy(), np.abs(val.numpy()), m, n, 'add' if op == 'sum' else op)
            assert np.all(np.abs(got.cpu().numpy() - ev) <= 1e-5 * l1 + 1e-30)
    assert saw_fused == fused_expected, (saw_fused, fused_expected)
    # index only (no value): the same pairs and counts, nothing reduced
    index_u, seg, counts, _, _ = ops.sort_coalesce_reduce(rd, cd, m, n, None, 0)
    assert counts.tolist() == counts_0.tolist() + [0]
    assert torch.equal(index_u[:, :k], index_0[:, :k])


def test_rank_self_test_and_forced_ballot_ranking(dev, ops):
    """The stable rank of the radix kernels is a returning LDS atomic when the device-side self-test finds the lanes of
    one instruction served in ascending order, ballot matching otherwise.  Both must give the SAME permutation on
    inputs made of a few hot keys (many lanes of one instruction on one counter)."""
    decided = ops.sort_rank_mode(2)  # run the self-test again
    assert decided in (0, 1)
    g = torch.Generator().manual_seed(11)
    cases = []
    for (E, m, n) in ((300000, 7, 9), (200000, 1 << 20, 1 << 20), (9000, 40, 3), (150000, 1 << 31, 1 << 30)):
        row = torch.randint(0, m, (E, ), generator=g)
        col = torch.randint(0, n, (E, ), generator=g)
        hot = torch.rand(E, generator=g) < 0.6
        row[hot], col[hot] = row[0], col[0]
        cases.append((row, col, m, n))
    try:
        outs = {}
        for mode in (0, 1):
            assert ops.sort_rank_mode(mode) == mode
            outs[mode] = [ops.sort_coo(r.to(dev), c.to(dev), m, n, True) for (r, c, m, n) in cases]
        for (r, c, m, n), a, b in zip(cases, outs[0], outs[1]):
            ep = no.sort_coo(r.numpy(), c.numpy(), m, n)[2]
            assert np.array_equal(b[2].cpu().numpy(), ep), 'ballot ranking'
            if decided == 0:
                assert np.array_equal(a[2].cpu().numpy(), ep), 'atomic ranking'
    finally:
        ops.sort_rank_mode(2)
    assert ops.sort_rank_mode(-1) == decided


def test_fused_coalesce_under_both_rankings(dev, ops):
    """The compacting bucket sort with the fused reduction has a ballot-ranking instantiation too: distinct pairs,
    counts and reduced values must not depend on the ranking in force (fp32 and int32, every reduction, index only)."""
    decided = ops.sort_rank_mode(-1)
    g = torch.Generator().manual_seed(23)
    cases = []
    for (E, m, n) in ((400003, 700, 900), (600000, 300000, 300000), (1 << 20, 16, 16)):
        row = torch.randint(0, m, (E, ), generator=g)
        col = torch.randint(0, n, (E, ), generator=g)
        q = E // 4
        row[:q], col[:q] = row[q:2 * q].clone(), col[q:2 * q].clone()
        cases.append((row.to(dev), col.to(dev), m, n, torch.randn(E, generator=g).to(dev),
                      torch.randint(-50, 50, (E, ), generator=g, dtype=torch.int32).to(dev)))
    try:
        outs = {}
        for mode in (0, 1):
            assert ops.sort_rank_mode(mode) == mode
            res = []
            for (r, c, m, n, vf, vi) in cases:
                for v in (vf, vi, None):
                    for code in ((0, 1, 2, 3) if v is not None else (0, )):
                        index_u, seg, counts, vs, vu = ops.sort_coalesce_reduce(r, c, m, n, v, code)
                        cl = counts.tolist()
                        k = cl[2]
                        res.append((cl, index_u[:, :k].clone(), vu[:k].clone() if (v is not None and cl[3] & 1) else None))
            outs[mode] = res
        saw_fused = False
        for a, b in zip(outs[0], outs[1]):
            assert a[0] == b[0] and torch.equal(a[1], b[1])
            assert (a[2] is None) == (b[2] is None)
            if a[2] is not None:
                saw_fused = True
                assert torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
        assert saw_fused
    finally:
        ops.sort_rank_mode(2)
    assert ops.sort_rank_mode(-1) == decided


def test_device_decided_sort_and_probe(dev, ops):
    E, m, n = 300000, 4000, 5000
    g = torch.Generator().manual_seed(2)
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    er, ec, ep = no.sort_coo(row.numpy(), col.numpy(), m, n)
    key = row.numpy().astype(np.int64) * n + col.numpy()
    # unsorted input: probe counts + the sorted outputs
    rs, cs, perm, counts = ops.sort_coo_auto(row.to(dev), col.to(dev), m, n)
    assert counts.tolist() == [int((key[1:] < key[:-1]).sum()), int((key[1:] == key[:-1]).sum())]
    assert np.array_equal(perm.cpu().numpy(), ep) and np.array_equal(rs.cpu().numpy(), er) and np.array_equal(cs.cpu().numpy(), ec)
    # sorted input (with duplicates): nothing is sorted, the outputs are a copy and the identity
    srow, scol = torch.from_numpy(er), torch.from_numpy(ec)
    rs, cs, perm, counts = ops.sort_coo_auto(srow.to(dev), scol.to(dev), m, n)
    skey = er.astype(np.int64) * n + ec
    assert counts.tolist() == [0, int((skey[1:] == skey[:-1]).sum())]
    assert torch.equal(perm.cpu(), torch.arange(E)) and torch.equal(rs.cpu(), srow) and torch.equal(cs.cpu(), scol)
    # probed variant: the count comes from coo_check
    for r_, c_, want_p in ((row, col, ep), (srow, scol, np.arange(E))):
        chk = ops.coo_check(r_.to(dev), c_.to(dev))
        rs, cs, perm = ops.sort_coo_probed(r_.to(dev), c_.to(dev), m, n, chk)
        assert np.array_equal(perm.cpu().numpy(), want_p) and np.array_equal(rs.cpu().numpy(), er)
    # negative ids read as huge maxima (the constructor's range assert then fires)
    bad = row.clone()
    bad[12345] = -1
    chk = ops.coo_check(bad.to(dev), col.to(dev)).tolist()
    assert chk[2] < 0 or chk[2] >= m


@pytest.mark.parametrize('E', [1, 2047, 2048, 2049, 500000, 3000001])
def test_fused_compaction(dev, ops, E):
    g = torch.Generator().manual_seed(E)
    m, n = 3000, 700
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    er, ec, _ = no.sort_coo(row.numpy(), col.numpy(), m, n)
    key = er.astype(np.int64) * n + ec
    head = np.ones(E, dtype=bool)
    head[1:] = key[1:] != key[:-1]
    pos = np.nonzero(head)[0]
    ru, cu, seg, nnz = ops.coalesce_index(torch.from_numpy(er).to(dev), torch.from_numpy(ec).to(dev))
    k = int(nnz)
    assert k == pos.size
    assert np.array_equal(ru[:k].cpu().numpy(), er[pos]) and np.array_equal(cu[:k].cpu().numpy(), ec[pos])
    assert np.array_equal(seg[:k + 1].cpu().numpy(), np.append(pos, E))


@pytest.mark.parametrize('E', [500, 20000])
def test_constructor_rejects_out_of_range_and_negative_ids(dev, E):
    """The enqueue-before-check constructor sorts an untrusted COO before its range assert fires (memory-safe:
    positions come from ranks); the assert must still fire -- for ids >= the size and for negative ids (read as
    unsigned maxima) -- on the one-launch path (E <= 8192) and on the general path."""
    import pytorch_sparse_amd as ts
    g = torch.Generator().manual_seed(E)
    m, n = 300, 200
    row = torch.randint(0, m, (E, ), generator=g)
    col = torch.randint(0, n, (E, ), generator=g)
    val = torch.rand(E, generator=g)
    A = ts.SparseTensor(row=row.to(dev), col=col.to(dev), value=val.to(dev), sparse_sizes=(m, n))
    er, ec, ep = no.sort_coo(row.numpy(), col.numpy(), m, n)
    r2, c2, v2 = A.coo()
    assert np.array_equal(r2.cpu().numpy(), er) and np.array_equal(c2.cpu().numpy(), ec)
    assert np.array_equal(v2.cpu().numpy(), val.numpy()[ep])
    assert np.array_equal(A.storage.rowptr().cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(er, minlength=m))]))
    for which, bad in (('row', m), ('row', -1), ('col', n), ('col', -5)):
        r_, c_ = row.clone(), col.clone()
        (r_ if which == 'row' else c_)[E // 2] = bad
        with pytest.raises((AssertionError, RuntimeError)):
            ts.SparseTensor(row=r_.to(dev), col=c_.to(dev), value=val.to(dev), sparse_sizes=(m, n))


def test_ind2ptr_never_writes_past_its_output_on_invalid_ids(dev):
    """the sort-on-construct path enqueues ind2ptr on rows whose range check has not been read back yet.
    Ids in (M, M + 1024) and negative ids must not make the fill loop write outside the (M + 1)-word output: the
    words in front of and behind it keep their sentinel."""
    from pytorch_sparse_amd import _native as nat
    L = nat.lib()
    M, guard = 1000, 4096
    for bad in ([M + 1, M + 500, M + 1023], [-3, -1], [M + 5000], [M, M]):
        ind = torch.tensor(sorted([5, 7, 7, 400] + bad), dtype=torch.long, device=dev)
        buf = torch.full((guard + M + 1 + guard, ), 0x5A5A5A5A, dtype=torch.long, device=dev)
        out = buf[guard:guard + M + 1]
        with torch.cuda.device(dev):
            st = L.tsamd_ind2ptr(nat._ptr(ind), nat._i64(M), nat._i64(ind.numel()), nat._ptr(out), nat.stream_ptr(dev))
        assert st == 0
        torch.cuda.synchronize()
        assert bool((buf[:guard] == 0x5A5A5A5A).all()) and bool((buf[guard + M + 1:] == 0x5A5A5A5A).all()), bad

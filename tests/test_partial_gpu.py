"""tsamd_spmm_partial (column-block partial products combined into one result) and the overlapped all-gather plan
built on it (pytorch_sparse_amd/parallel.py).  Checker: the unsharded HIP product (itself pinned to the
oracle by tests/test_spmm_gpu.py) and tests/util.ref_partial (the contract restated on the C oracle)."""
import pytest
import torch

from pytorch_sparse_amd import synth
from tests.util import SUM_TOL, bits_equal, ref_partial

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nat():
    from pytorch_sparse_amd import _native
    _native.lib()
    return _native


def _split_blocks(rp, c, nblocks, seed):
    """Assign every entry to one of `nblocks` column blocks (by a random map of the column ids) -> per block
    (rowptr over all rows, col, src)."""
    M = rp.numel() - 1
    g = torch.Generator().manual_seed(seed)
    n = int(c.max()) + 1 if c.numel() else 1
    blk_of_col = torch.randint(0, nblocks, (n, ), generator=g).to(c.device)
    row = torch.repeat_interleave(torch.arange(M, device=c.device), rp[1:] - rp[:-1])
    out = []
    for b in range(nblocks):
        src = torch.nonzero(blk_of_col[c] == b).view(-1) if c.numel() else c.new_zeros(0)
        counts = torch.bincount(row[src], minlength=M) if src.numel() else torch.zeros(M, dtype=torch.long, device=c.device)
        brp = torch.zeros(M + 1, dtype=torch.long, device=c.device)
        torch.cumsum(counts, 0, out=brp[1:])
        out.append((brp, c[src].contiguous(), src))
    return out


def _run_staged(nat, blocks, order, v, x, reduce, full_rp, E, fn=None):
    M, K = full_rp.numel() - 1, x.size(1)
    out = torch.full((M, K), 7, dtype=x.dtype, device=x.device)  # garbage: the first block must overwrite it
    arg = torch.full((M, K), -3, dtype=torch.long, device=x.device) if reduce in ('min', 'max') else None
    for i, b in enumerate(order):
        brp, bc, src = blocks[b]
        red = reduce
        if reduce == 'mean' and i < len(order) - 1:
            red = 'sum'
        bv = None if v is None else v[src]
        args = (brp, bc, bv, x, red, out, arg, src if arg is not None else None, E, i > 0,
                full_rp if red == 'mean' else None)
        if fn is None:
            nat.spmm_partial(*args[:5], out, arg, arg_map=args[7], arg_none=E, accumulate=i > 0, deg_rowptr=args[10])
        else:
            fn(*args)
    return out, arg


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('K', [4, 32, 128, 136])
def test_partial_products_match_the_whole_product(dev, nat, dtype, K):
    rp, c = synth.rmat_csr(11, 12, seed=3)
    n, E = rp.numel() - 1, c.numel()
    rp, c = rp.to(dev), c.to(dev)
    for has_value in (True, False):
        v = synth.values(E, dtype=dtype, device=dev) if has_value else None
        # small integers: plenty of exact ties between entries of different blocks (min / max arg rule)
        x = torch.randint(-3, 4, (n, K), generator=torch.Generator().manual_seed(5)).to(dtype).to(dev)
        if has_value:
            v = torch.randint(1, 3, (E, ), generator=torch.Generator().manual_seed(6)).to(dtype).to(dev)
        blocks = _split_blocks(rp, c, 4, seed=1)
        for reduce in ('sum', 'mean', 'min', 'max'):
            want, warg = nat.spmm(rp, c, v, x, reduce)
            for order in ([0, 1, 2, 3], [3, 1, 0, 2]):
                got, garg = _run_staged(nat, blocks, order, v, x, reduce, rp, E)
                if reduce in ('min', 'max'):
                    assert bits_equal(got, want) and torch.equal(garg, warg), (dtype, K, reduce, order, has_value)
                elif reduce == 'sum' and dtype in (torch.float32, torch.float64):
                    assert bits_equal(got, want), (dtype, K, reduce)  # integer-valued: every sum is exact
                else:
                    tol = SUM_TOL[dtype]
                    assert torch.allclose(got.double(), want.double(), rtol=4 * tol, atol=4 * tol), (dtype, K, reduce)


def test_partial_against_the_restated_contract_and_edge_cases(dev, nat):
    """Random real-valued operands: block after block against tests/util.ref_partial on the host (min / max state
    bit for bit after EVERY block; sums to rounding); rows without entries, rows whose entries are all NaN, an empty
    block, a batch dimension."""
    rp, c = synth.rmat_csr(9, 6, seed=8)
    n, E = rp.numel() - 1, c.numel()
    K = 20
    x = synth.features(n, K, seed=2)
    v = synth.values(E, seed=3)
    x[5] = float('nan')  # column 5 is a hub: many rows see a NaN candidate (never wins)
    blocks = _split_blocks(rp, c, 3, seed=4)
    blocks.append((torch.zeros(n + 1, dtype=torch.long), c.new_zeros(0), c.new_zeros(0)))  # an empty block
    for reduce in ('min', 'max', 'sum', 'mean'):
        order = [3, 0, 1, 2] if reduce != 'mean' else [0, 3, 1, 2]
        M = n
        out_g = torch.empty(M, K, device=dev)
        arg_g = torch.empty(M, K, dtype=torch.long, device=dev) if reduce in ('min', 'max') else None
        out_c = torch.empty(M, K)
        arg_c = torch.empty(M, K, dtype=torch.long) if reduce in ('min', 'max') else None
        for i, b in enumerate(order):
            brp, bc, src = blocks[b]
            red = 'sum' if (reduce == 'mean' and i < len(order) - 1) else reduce
            bv = v[src]
            nat.spmm_partial(brp.to(dev), bc.to(dev), bv.to(dev), x.to(dev), red, out_g, arg_g,
                             arg_map=src.to(dev) if arg_g is not None else None, arg_none=E, accumulate=i > 0,
                             deg_rowptr=rp.to(dev) if red == 'mean' else None)
            ref_partial(brp, bc, bv, x, red, out_c, arg_c, src if arg_c is not None else None, E, i > 0,
                        rp if red == 'mean' else None)
            if arg_g is not None:
                assert torch.equal(arg_g.cpu(), arg_c), (reduce, i)
                assert bits_equal(out_g, out_c), (reduce, i)
            else:
                fin = torch.isfinite(out_c)
                assert torch.equal(torch.isfinite(out_g.cpu()), fin)
                assert torch.allclose(out_g.cpu()[fin], out_c[fin], rtol=1e-5, atol=1e-5), (reduce, i)
    # batch dimension: [B, N, K]
    xb = synth.features(2 * n, 8, seed=9).view(2, n, 8).to(dev)
    xb = torch.nan_to_num(xb)
    want, warg = nat.spmm(rp.to(dev), c.to(dev), v.to(dev), xb, 'max')
    out = torch.empty(2, n, 8, device=dev)
    arg = torch.empty(2, n, 8, dtype=torch.long, device=dev)
    for i, (brp, bc, src) in enumerate(blocks[:3]):
        nat.spmm_partial(brp.to(dev), bc.to(dev), v[src].to(dev), xb, 'max', out, arg, arg_map=src.to(dev), arg_none=E,
                         accumulate=i > 0)
    assert bits_equal(out, want) and torch.equal(arg, warg)
    # integer types have no partial products (feature matrices only)
    with pytest.raises(nat.TsamdError):
        xi = torch.ones(n, 4, dtype=torch.int32, device=dev)
        nat.spmm_partial(rp.to(dev), c.to(dev), None, xi, 'sum', torch.empty(n, 4, dtype=torch.int32, device=dev),
                         accumulate=False)


@pytest.mark.parametrize('P,chunks', [(2, 1), (4, 3), (8, 4)])
def test_overlapped_allgather_plan_with_logical_ranks(dev, nat, P, chunks):
    """The column stages of OverlappedAllGatherSpMM for every one of P logical ranks on this device: the landing
    buffers are filled the way the chunk collectives fill them (rank p's chunk c in rows [p cs, (p + 1) cs) of
    buffer c, in the hashed wire order), the stages run through tsamd_spmm_partial, and the stacked result equals
    the unsharded product (max: bit for bit incl. arg ids; sum / mean: 1e-5)."""
    from pytorch_sparse_amd.parallel import _default_positions, build_column_stages, narrow_rows, partition_rows
    rp, c = synth.rmat_csr(13, 16, seed=4, device=dev)
    n, E, K = rp.numel() - 1, c.numel(), 64
    v = synth.values(E, device=dev)
    x = synth.features(n, K, device=dev)
    ranges = partition_rows(rp, P, 'nnz')
    x_sizes = [e - s for s, e in ranges]
    positions = [_default_positions(k, dev) for k in x_sizes]
    assert all(torch.equal(torch.sort(p_)[0], torch.arange(k, device=dev)) for p_, k in zip(positions, x_sizes))
    cs = None
    # every rank's shard in wire order, padded to chunks * cs rows
    full = {red: nat.spmm(rp, c, v, x, red) for red in ('sum', 'mean', 'max', 'min')}
    for rank, (s, e) in enumerate(ranges):
        lrp, lc, lv = narrow_rows(rp, c, v, s, e)
        lrp = lrp.contiguous()
        cs, stages = build_column_stages(lrp, lc, x_sizes, rank, chunks, positions)
        assert sum(st['src'].numel() for st in stages) == lc.numel()
        pads = []
        for p_, (ps, pe) in enumerate(ranges):
            xp = torch.zeros(chunks * cs, K, device=dev)
            xp[positions[p_]] = x[ps:pe]
            pads.append(xp)
        bufs = [torch.cat([pads[p_][ch * cs:(ch + 1) * cs] for p_ in range(P)]) for ch in range(chunks)]
        e0 = int(rp[s])
        for reduce in ('sum', 'mean', 'max', 'min'):
            minmax = reduce in ('min', 'max')
            out = torch.empty(e - s, K, device=dev)
            arg = torch.empty(e - s, K, dtype=torch.long, device=dev) if minmax else None
            for i, st in enumerate(stages):
                red = 'sum' if (reduce == 'mean' and i < len(stages) - 1) else reduce
                nat.spmm_partial(st['rowptr'], st['col'], lv[st['src']], pads[rank] if i == 0 else bufs[i - 1], red, out, arg,
                                 arg_map=st['src'] if minmax else None, arg_none=lc.numel(), accumulate=i > 0,
                                 deg_rowptr=lrp if red == 'mean' else None)
            want, warg = full[reduce]
            if minmax:
                wl = torch.where(warg[s:e] == E, torch.full_like(warg[s:e], lc.numel()), warg[s:e] - e0)
                assert bits_equal(out, want[s:e]) and torch.equal(arg, wl), (rank, reduce)
            else:
                assert torch.allclose(out, want[s:e], rtol=1e-5, atol=1e-5), (rank, reduce)

"""One-off differential fuzzer for the SpMM forward / backward kernels against the C oracle at medium
sizes (the in-suite fuzz test covers tiny shapes).  Not collected by pytest; lives under tests/
because it drives the oracle.  Usage (on a GPU box):

    python tests/fuzz_spmm.py [--cases 400] [--seed 0]

Draws: M up to 2e5 rows, degree laws (uniform / Zipf / a few hubs / block-empty), N up to 1e6
columns incl. hub-heavy column laws that trigger the relabel probe, K in 1..300, all six dtypes, four
reductions, batches, TSAMD_SPMM_RELABEL in {auto, 0, 1}.  Prints one line per failure and a summary;
exit code 1 when anything failed."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import c_oracle as oc  # noqa: E402
from pytorch_sparse_amd import _native as nat  # noqa: E402
from tests.util import SUM_ATOL, SUM_TOL, check_spmm  # noqa: E402

DTS = [torch.float32, torch.float64, torch.bfloat16, torch.float16, torch.int32, torch.int64]


def degrees(rng, M):
    kind = rng.integers(5)
    if kind == 0:
        return rng.integers(0, 12, M)
    if kind == 1:  # Zipf
        return np.minimum(rng.zipf(1.7, M) - 1, 50_000)
    if kind == 2:  # a few hubs
        d = rng.integers(0, 3, M)
        d[rng.integers(0, M, 3)] = rng.integers(10_000, 300_000, 3)
        return d
    if kind == 3:  # long empty stretches
        d = rng.integers(0, 40, M)
        d[rng.random(M) < 0.8] = 0
        return d
    return np.full(M, int(rng.choice([1, 63, 64, 65, 128])))


def columns(rng, E, N):
    kind = rng.integers(3)
    if kind == 0:
        return rng.integers(0, N, E)
    if kind == 1:  # hub columns that are multiples of a big power of two (channel camping pattern)
        step = 1 << int(rng.integers(3, 12))
        hubs = (rng.integers(0, max(N // step, 1), 64) * step) % N
        c = rng.integers(0, N, E)
        m = rng.random(E) < 0.6
        c[m] = hubs[rng.integers(0, 64, int(m.sum()))]
        return c
    return np.minimum(rng.zipf(1.5, E) - 1, N - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=400)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(args.seed)
    fails = 0
    for case in range(args.cases):
        M = int(rng.choice([1, 100, 5_000, 60_000, 200_000]))
        N = int(rng.choice([1, 50, 4_000, 100_000, 1_000_000]))
        deg = degrees(rng, M).astype(np.int64)
        budget = 3_000_000
        if deg.sum() > budget:
            deg = (deg * (budget / deg.sum())).astype(np.int64)
        rp = np.zeros(M + 1, np.int64)
        np.cumsum(deg, out=rp[1:])
        E = int(rp[-1])
        c = columns(rng, E, N).astype(np.int64)
        dtype = DTS[rng.integers(len(DTS))]
        K = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 64, 100, 128, 257, 300]))
        if E * K > 120_000_000:
            K = max(1, 120_000_000 // max(E, 1))
        batch = () if rng.random() < 0.8 else (int(rng.integers(1, 3)), )
        has_value = bool(rng.random() < 0.6)
        reduce = ['sum', 'mean', 'min', 'max'][rng.integers(4)]
        relabel = ['auto', '0', '1'][rng.integers(3)]
        os.environ['TSAMD_SPMM_RELABEL'] = relabel
        g = torch.Generator().manual_seed(case)
        if dtype.is_floating_point:
            v = (torch.rand(E, generator=g) - 0.3).to(dtype) if has_value else None
            x = (torch.rand((*batch, N, K), generator=g) - 0.5).to(dtype)
        else:
            v = torch.randint(-4, 5, (E, ), dtype=dtype, generator=g) if has_value else None
            x = torch.randint(-9, 9, (*batch, N, K), dtype=dtype, generator=g)
        rpt, ct = torch.from_numpy(rp), torch.from_numpy(c)
        tag = 'case %d: M=%d N=%d E=%d K=%d %s %s batch=%s value=%s relabel=%s' % (
            case, M, N, E, K, dtype, reduce, batch, has_value, relabel)
        try:
            out, arg = nat.spmm(rpt.to(dev), ct.to(dev), None if v is None else v.to(dev), x.to(dev), reduce)
            torch.cuda.synchronize()
            check_spmm(out, arg, rpt, ct, v, x, reduce)
            if dtype.is_floating_point and E > 0 and case % 3 == 0:
                gout = (torch.rand((*batch, M, K), generator=g) - 0.5).to(dtype)
                if reduce in ('sum', 'mean'):
                    row = torch.from_numpy(oc.ptr2ind(rp, E))
                    got = nat.spmm_value_bw(row.to(dev), rpt.to(dev), ct.to(dev), x.to(dev), gout.to(dev), reduce)
                    ex = oc.spmm_value_bw(oc.F64, reduce, row.numpy(), rp, c, x.double().numpy(), gout.double().numpy())
                    l1 = oc.spmm_value_bw(oc.F64, reduce, row.numpy(), rp, c, x.double().abs().numpy(),
                                          gout.double().abs().numpy())
                    err = np.abs(got.cpu().double().numpy() - ex)
                    assert (err <= SUM_TOL[dtype] * l1 + SUM_ATOL[dtype]).all(), 'value_bw'
                else:
                    gv, gm = nat.spmm_minmax_bw(rpt.to(dev), ct.to(dev), None if v is None else v.to(dev), x.to(dev), gout.to(dev),
                                                arg, want_value=has_value, want_mat=True)
                    egv, egm = oc.spmm_minmax_bw(oc.F64, c, None if v is None else v.double().numpy(),
                                                 x.double().numpy(), gout.double().numpy(), arg.cpu().numpy(),
                                                 want_value=has_value)
                    tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
                    # grad_mat is accumulated by hardware atomics in the element type, one rounding per
                    # addition in any order (like the reference's scatter_add_): (#addends + 1) * u * sum|terms|
                    absv = None if v is None else v.double().abs().numpy()
                    _, l1 = oc.spmm_minmax_bw(oc.F64, c, absv, x.double().numpy(), gout.double().abs().numpy(),
                                              arg.cpu().numpy(), want_value=False)
                    _, cnt = oc.spmm_minmax_bw(oc.F64, c, None, x.double().numpy(), np.ones_like(gout.double().numpy()),
                                               arg.cpu().numpy(), want_value=False)
                    u = {torch.float32: 2.0 ** -24, torch.float64: 2.0 ** -53, torch.float16: 2.0 ** -11,
                         torch.bfloat16: 2.0 ** -8}[dtype]
                    err = np.abs(gm.cpu().double().numpy() - egm)
                    # (+ half a subnormal step per addition: fp16 results below 6e-5 are not relatively accurate)
                    floor = 2.0 ** -25 if dtype == torch.float16 else 1e-40
                    assert (err <= (cnt + 1) * (u * l1 * 1.01 + floor)).all(), 'minmax_bw mat'
                    if has_value:
                        s = max(1.0, float(np.abs(egv).max()))
                        assert np.allclose(gv.cpu().double().numpy(), egv, rtol=tol, atol=tol * s), 'minmax_bw value'
        except Exception as exc:  # noqa: BLE001
            fails += 1
            print('FAIL', tag, '::', type(exc).__name__, str(exc)[:200], flush=True)
        if case % 50 == 49:
            print('... %d cases, %d failures' % (case + 1, fails), flush=True)
    print('fuzz: %d cases, %d failures (seed %d)' % (args.cases, fails, args.seed), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()

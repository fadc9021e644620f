"""Shared helpers for the parity tests: torch <-> oracle (numpy) plumbing and tolerances."""
import numpy as np
import torch

from oracle import c_oracle as oc

CODE = {torch.float32: oc.F32, torch.float64: oc.F64, torch.float16: oc.F16,
        torch.bfloat16: oc.BF16, torch.int32: oc.I32, torch.int64: oc.I64, torch.uint8: oc.U8,
        torch.int8: oc.I8, torch.int16: oc.I16}
ALL_DTYPES = list(CODE)
FLOAT_DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16]
# |gpu - exact| <= TOL * sum_e |value_e * x_e|   (fp32: the 1e-5 rel bar of BASELINE.json;
# narrow types: a few ulp of the once-rounded result)
SUM_TOL = {torch.float32: 1e-5, torch.float64: 1e-13, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}
# absolute floor: one fp16 subnormal step (results below 6e-5 cannot be relatively accurate)
SUM_ATOL = {torch.float32: 1e-37, torch.float64: 1e-300, torch.float16: 6e-8, torch.bfloat16: 1e-37}


def experiments_build():
    """True when the loaded libtsamd.so was built with -DTSAMD_EXPERIMENTS=1 (scripts/variants.py): only then do the
    TSAMD_* environment switches select the alternative kernel variants (include/tsamd.h: tsamd_build_flags)."""
    from pytorch_sparse_amd import _native as nat
    return bool(nat.lib().tsamd_build_flags() & 1)


def tonp(t):
    if t is None:
        return None
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def fromnp(a, dtype):
    if dtype == torch.bfloat16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def bits_equal(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype in (torch.bfloat16, torch.float16):
        return torch.equal(a.view(torch.int16), b.view(torch.int16))
    if a.dtype == torch.float32:
        return torch.equal(a.view(torch.int32), b.view(torch.int32))
    if a.dtype == torch.float64:
        return torch.equal(a.view(torch.int64), b.view(torch.int64))
    return torch.equal(a, b)


def oracle_spmm(rowptr, col, value, mat, reduce, wide_acc=False):
    """C oracle on torch CPU tensors; returns torch tensors (out, arg)."""
    o, a = oc.spmm(CODE[mat.dtype], reduce, tonp(rowptr), tonp(col), tonp(value), tonp(mat),
                   wide_acc=wide_acc)
    return fromnp(o, mat.dtype), (None if a is None else torch.from_numpy(a))


def exact_sum_and_l1(rowptr, col, value, mat, reduce):
    """fp64 result of sum/mean and the per-output L1 mass bound used for tolerances."""
    rp, c = tonp(rowptr), tonp(col)
    v64 = None if value is None else value.detach().cpu().double().numpy()
    x64 = mat.detach().cpu().double().numpy()
    ex, _ = oc.spmm(oc.F64, reduce, rp, c, v64, x64)
    l1, _ = oc.spmm(oc.F64, reduce, rp, c, None if v64 is None else np.abs(v64), np.abs(x64))
    return ex, l1


def check_spmm(out, arg, rowptr, col, value, mat, reduce):
    """Assert the GPU result (out, arg) against the oracle.  min/max and integer work are
    bit-exact; floating sum/mean are bounded by SUM_TOL * L1 mass (fp32: 1e-5)."""
    dtype = mat.dtype
    rowptr, col = rowptr.cpu(), col.cpu()
    value = None if value is None else value.cpu()
    mat = mat.cpu()
    if reduce in ('min', 'max') or not dtype.is_floating_point:
        eo, ea = oracle_spmm(rowptr, col, value, mat, reduce)
        assert bits_equal(out, eo), 'value mismatch (%s, %s)' % (dtype, reduce)
        if ea is not None:
            assert torch.equal(arg.cpu(), ea), 'arg_out mismatch (%s, %s)' % (dtype, reduce)
    else:
        ex, l1 = exact_sum_and_l1(rowptr, col, value, mat, reduce)
        got = out.detach().cpu().double().numpy()
        bound = SUM_TOL[dtype] * l1 + SUM_ATOL[dtype]
        if dtype == torch.float16:  # results beyond the fp16 range must overflow to +-inf, not be "close"
            big = np.abs(ex) > 65504.0 * (1 + 2.0 ** -11)
            assert (np.isinf(got[big]) & (np.sign(got[big]) == np.sign(ex[big]))).all(), 'fp16 overflow'
            near = np.abs(ex) > 65504.0 * (1 - 2.0 ** -9)  # within rounding distance of the limit: either way
            got, ex, bound = got[~big & ~near], ex[~big & ~near], bound[~big & ~near]
        err = np.abs(got - ex)
        assert (err <= bound).all(), 'max err/bound %.3g (%s, %s)' % ((err / bound).max(), dtype, reduce)


def ref_partial(rowptr, col, value, mat, reduce, out, arg_out, arg_map, arg_none, accumulate, deg_rowptr):
    """Restatement of tsamd_spmm_partial's contract (include/tsamd.h) on CPU tensors: the block product by the C
    oracle, then the combine rule -- sums add, min / max keep the better candidate with ties to the smaller entry id
    of the WHOLE matrix (csrc/cpu/reducer.h:63-67 applied across blocks); `arg_none` marks "no winner so far".
    Writes `out` / `arg_out` in place.  Used as the injected partial_fn of the gloo tests and as the checker of
    the HIP kernel on the GPU."""
    E = col.numel()
    red = 'sum' if reduce == 'mean' else reduce
    po, pa = oracle_spmm(rowptr, col, value, mat, red)
    if reduce in ('sum', 'mean'):
        tot = po if not accumulate else (out.to(torch.float64) + po.to(torch.float64)).to(out.dtype) \
            if out.dtype != torch.float32 else out + po
        if reduce == 'mean':
            deg = (deg_rowptr[1:] - deg_rowptr[:-1]).clamp(min=1).to(tot.dtype)
            tot = tot / deg.view(-1, *([1] * (tot.dim() - 1)))
        out.copy_(tot)
        return
    deg = (rowptr[1:] - rowptr[:-1]).view(-1, *([1] * (po.dim() - 1))).expand_as(po) if po.dim() == 2 else \
        (rowptr[1:] - rowptr[:-1]).view(1, -1, 1).expand_as(po)
    none = pa == E
    if E == 0:
        ca = torch.full_like(pa, arg_none)
    else:
        ca = torch.where(none, torch.full_like(pa, arg_none), pa if arg_map is None else arg_map[pa.clamp(max=E - 1)])
    if not accumulate:
        out.copy_(po)
        arg_out.copy_(ca)
        return
    ev, ea = out.clone(), arg_out.clone()
    pv, evv = po.double(), ev.double()
    better = (pv < evv) if reduce == 'min' else (pv > evv)
    take = torch.where(ea == arg_none, torch.ones_like(none), torch.where(ca == arg_none, torch.zeros_like(none),
                                                                       better | ((pv == evv) & (ca < ea))))
    take = take & (deg > 0)
    out.copy_(torch.where(take, po, ev))
    arg_out.copy_(torch.where(take, ca, ea))


def ref_value_bw(row, rowptr, col, mat, grad):
    """Injected value_bw_fn of the gloo tests: the C oracle's SDDMM (csrc/cpu/spmm_cpu.cpp:103-152 restated)."""
    gv = oc.spmm_value_bw(CODE[mat.dtype], 'sum', tonp(row), tonp(rowptr), tonp(col), tonp(mat), tonp(grad))
    return fromnp(gv, mat.dtype)


def ref_minmax_bw(rowptr, col, value, mat, grad, arg, want_value):
    """Injected minmax_bw_fn of the gloo tests: the C oracle's restatement of csrc/spmm.cpp:204-242 -> (grad_value or
    None, grad_mat)."""
    gv, gm = oc.spmm_minmax_bw(CODE[mat.dtype], tonp(col), tonp(value), tonp(mat), tonp(grad), tonp(arg),
                               want_value=want_value and value is not None, want_mat=True)
    return (None if gv is None else fromnp(gv, mat.dtype)), fromnp(gm, mat.dtype)

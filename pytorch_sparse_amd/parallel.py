"""1-D row-sharded SpMM across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; its slicing primitive ``narrow(src, 0, start, length)``
(torch_sparse/narrow.py:15-42) defines what a row shard is: ``rowptr[start:start+length+1] -
rowptr[start]`` with the matching slices of ``col`` / ``value`` and *global* column ids.

Path per step (BASELINE.json north_star), class ``RowShardedSpMM`` (the plain, serial form):
    X_full = all_gather(X_local)           RCCL, the only data-path collective
    out_local = A_local @ X_full           tsamd_spmm on the local row block
``OverlappedAllGatherSpMM`` is the same all-gather hidden behind the compute: the local row block
is pre-split into column blocks, X travels in row chunks (already in the channel-camping-free row order, so the
gathered operand is never copied again), and the partial product of a block starts as soon as its rows landed
(tsamd_spmm_partial accumulates into the result; the local block needs no exchange and runs first).
``HaloShardedSpMM`` is the same computation with the collective reduced to the rows of X the local
block references (all_to_all_single with uneven splits, planned once per matrix).
The output stays row-sharded, so there is no reduce step in the forward.  In the backward the
gradient of X is a partial sum on every rank and is reduce-scattered to its owners; the gradient
of the sparse values is local.

``spmm_fn`` is injectable so that the sharding / collective logic can be exercised on CPU with the
gloo backend (tests/test_parallel_cpu.py); the product default is the HIP operator.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def partition_rows(rowptr: Tensor, parts: int, balance: str = 'nnz') -> List[Tuple[int, int]]:
    """Contiguous row ranges [(start, end)] for `parts` ranks.

    balance='rows': equal row counts.  balance='nnz': split points at equal shares of the
    non-zeros (searchsorted on rowptr) -- what a power-law matrix needs."""
    M = rowptr.numel() - 1
    if balance == 'rows':
        cuts = [(M * p) // parts for p in range(parts + 1)]
    elif balance == 'nnz':
        E = int(rowptr[-1])
        targets = torch.tensor([(E * p) // parts for p in range(1, parts)], dtype=rowptr.dtype,
                               device=rowptr.device)
        inner = torch.searchsorted(rowptr, targets, right=False).clamp_(0, M).tolist() if parts > 1 else []
        cuts = [0] + inner + [M]
        for i in range(1, len(cuts)):  # monotone even with empty stretches
            cuts[i] = max(cuts[i], cuts[i - 1])
    else:
        raise ValueError(balance)
    return [(cuts[p], cuts[p + 1]) for p in range(parts)]


def narrow_rows(rowptr: Tensor, col: Tensor, value: Optional[Tensor], start: int, end: int):
    """Row block [start, end) as a local CSR with global column ids (views, no copies of col/value)."""
    e0, e1 = int(rowptr[start]), int(rowptr[end])
    return rowptr[start:end + 1] - e0, col[e0:e1], None if value is None else value[e0:e1]


def pack_rows(x: Tensor, idx: Tensor) -> Tensor:
    """x[idx] for a send buffer: the hand-written gather (tsamd::gather_rows, one 16-byte packet per
    lane) on the GPU -- outside autograd, the exchange Functions own the gradient -- else ATen."""
    if x.is_cuda and x.dim() == 2 and not x.requires_grad:
        return torch.ops.tsamd.gather_rows(x, idx)
    return x.index_select(0, idx)


def _default_spmm_into(rowptr, col, value, x, reduce, out):
    """The local product written straight into `out` (a row slice of the rank's result): the C-ABI takes
    caller-allocated outputs, so the pieces of a pipelined step need no concatenation afterwards."""
    from . import _native as nat
    if value is not None and value.dtype != x.dtype:
        value = value.to(x.dtype)
    nat.spmm(rowptr, col, value, x, reduce, out=out)


def _default_spmm(rowptr, col, value, x, reduce):
    from .matmul import matmul
    from .tensor import SparseTensor
    A = SparseTensor(rowptr=rowptr, col=col, value=value, sparse_sizes=(rowptr.numel() - 1, x.size(-2)),
                     is_sorted=True, trust_data=True)
    return matmul(A, x, reduce)


class _GatherRows(torch.autograd.Function):
    """all_gather of row blocks of X; backward = reduce-scatter of the gradient to the owners."""

    @staticmethod
    def forward(ctx, x_local: Tensor, sizes: Sequence[int], group):
        ctx.sizes, ctx.group = list(sizes), group
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        ctx.rank = rank
        F = x_local.shape[1:]
        if len(set(sizes)) == 1:  # equal shards: one collective straight into the result
            out = x_local.new_empty((sum(sizes), ) + tuple(F))
            dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
            return out
        mx = max(sizes)
        pad = x_local.new_zeros((mx, ) + tuple(F))
        pad[:x_local.size(0)] = x_local
        buf = x_local.new_empty((world * mx, ) + tuple(F))
        dist.all_gather_into_tensor(buf, pad, group=group)
        return torch.cat([buf[r * mx:r * mx + sizes[r]] for r in range(world)], 0)

    @staticmethod
    def backward(ctx, grad_full: Tensor):
        sizes, group, rank = ctx.sizes, ctx.group, ctx.rank
        grad_full = grad_full.contiguous()
        start = sum(sizes[:rank])
        if len(set(sizes)) == 1 and dist.get_backend(group) != 'gloo':
            out = grad_full.new_empty((sizes[rank], ) + tuple(grad_full.shape[1:]))
            dist.reduce_scatter_tensor(out, grad_full, group=group)
            return out, None, None
        dist.all_reduce(grad_full, group=group)  # gloo has no reduce_scatter
        return grad_full[start:start + sizes[rank]].clone(), None, None


class RowShardedSpMM(object):
    """The local row block of A plus the bookkeeping to multiply it with a row-sharded X.

    rowptr/col/value: local CSR (global column ids) -- e.g. from ``narrow_rows``.
    x_sizes: number of X rows owned by each rank (sum = number of columns of A).
    """

    def __init__(self, rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                 group=None, spmm_fn: Optional[Callable] = None):
        self.rowptr, self.col, self.value = rowptr, col, value
        self.x_sizes = list(x_sizes)
        self.group = group
        self.spmm_fn = spmm_fn or _default_spmm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert len(self.x_sizes) == self.world

    def gather(self, x_local: Tensor) -> Tensor:
        if self.world == 1:
            return x_local
        return _GatherRows.apply(x_local, self.x_sizes, self.group)

    def __call__(self, x_local: Tensor, reduce: str = 'sum') -> Tensor:
        """out_local [rows of this rank, F] = A_local @ all_gather(x_local)."""
        return self.spmm_fn(self.rowptr, self.col, self.value, self.gather(x_local), reduce)


# ---------------------------------------------------------------------------------------------
# all-gather overlapped with per-column-block partial products
# ---------------------------------------------------------------------------------------------
def _default_positions(n: int, device) -> Tensor:
    """Row order of a shard on the wire: the hashed positions of the relabelled layout (a bijection of [0, n)
    that spreads the hub rows of Kronecker-like graphs over the memory channels) on the GPU,
    the identity elsewhere (CPU tests)."""
    if torch.device(device).type == 'cuda':
        like = torch.empty(0, dtype=torch.long, device=device)
        return torch.ops.tsamd.relabel_ids(None, n, like)
    return torch.arange(n, dtype=torch.long, device=device)


def build_column_stages(rowptr: Tensor, col: Tensor, x_sizes: Sequence[int], rank: int, chunks: int,
                        positions: Optional[Sequence[Tensor]] = None):
    """Split a local row block (global column ids) into the column blocks of the overlapped all-gather.

    Every owner p stores (and sends) its shard of X in the row order positions[p] (row i at position
    positions[p][i]; None = identity), padded to chunks * cs rows, cs = ceil(max shard / chunks).  Collective c
    gathers chunk c -- positions [c cs, (c + 1) cs) -- of every shard into a buffer [world * cs, K] whose row
    p * cs + j holds position c * cs + j of owner p.
      stage 0       the entries whose column this rank owns:   col = position in its own shard
      stage c + 1   the entries of the other owners' chunk c:  col = row of buffer c
    -> (cs, [dict(rowptr, col, src)]): src = ids of the stage's entries in the block's CSR (ascending: every stage
    is a sub-sequence of every row, so "first occurrence" ties are decided by src), rowptr over ALL local rows."""
    dev = col.device
    world = len(x_sizes)
    M = rowptr.numel() - 1
    mx = max(int(v) for v in x_sizes) if world > 0 else 0
    cs = max(1, -(-mx // max(1, chunks)))
    bounds = torch.tensor([0] + list(torch.tensor(list(x_sizes)).cumsum(0).tolist()), dtype=torch.long, device=dev)
    owner = torch.searchsorted(bounds, col, right=True) - 1
    local = col - bounds[owner]
    if positions is not None:
        pos = torch.empty_like(local)
        for p in range(world):
            sel = owner == p
            if positions[p] is not None and bool(sel.any()):
                pos[sel] = positions[p][local[sel]]
            else:
                pos[sel] = local[sel]
    else:
        pos = local
    chunk = torch.div(pos, cs, rounding_mode='floor')
    row = torch.repeat_interleave(torch.arange(M, dtype=torch.long, device=dev), rowptr[1:] - rowptr[:-1])
    stage_of = torch.where(owner == rank, torch.zeros_like(chunk), chunk + 1)
    stage_col = torch.where(owner == rank, pos, owner * cs + (pos - chunk * cs))
    stages = []
    for st in range(chunks + 1):
        src = torch.nonzero(stage_of == st).view(-1)
        counts = torch.bincount(row[src], minlength=M) if src.numel() > 0 else torch.zeros(M, dtype=torch.long, device=dev)
        rp = torch.zeros(M + 1, dtype=torch.long, device=dev)
        torch.cumsum(counts, 0, out=rp[1:])
        stages.append(dict(rowptr=rp, col=stage_col[src].contiguous(), src=src))
    return cs, stages


def _native_partial(rowptr, col, value, mat, reduce, out, arg_out, arg_map, arg_none, accumulate, deg_rowptr):
    from . import _native as nat
    if value is not None and value.dtype != mat.dtype:
        value = value.to(mat.dtype)
    nat.spmm_partial(rowptr, col, value, mat, reduce, out, arg_out, arg_map, arg_none, accumulate, deg_rowptr)


def _native_value_bw(row, rowptr, col, mat, grad):
    """[E] gradient of the block's entries (SDDMM): tsamd_spmm_value_bw."""
    from . import _native as nat
    return nat.spmm_value_bw(row, rowptr, col, mat, grad, 'sum')


def _native_minmax_bw(rowptr, col, value, mat, grad, arg, want_value):
    """(grad_value [E] or None, grad_mat [rows of mat, K]) of a min / max block from its winners: tsamd_spmm_minmax_bw."""
    from . import _native as nat
    if value is not None and value.dtype != mat.dtype:
        value = value.to(mat.dtype)
    return nat.spmm_minmax_bw(rowptr, col, value, mat, grad, arg, want_value=want_value, want_mat=True)


class _OverlappedProduct(torch.autograd.Function):
    """One training step of an ``OverlappedAllGatherSpMM`` as ONE autograd node (round 5; before, every stage went
    through the drop-in operator -- a probe and possibly a relabelled copy of each landed buffer per stage -- and
    min / max fell back to the serial exchange).

    forward  = the inference step: chunk collectives enqueued, partial products of the landed column blocks through
               ``partial_fn`` (tsamd_spmm_partial: no copy of the landed operand); the landed buffers, the result's
               winners (min / max) and the weights are kept for the backward.
    backward, sum / mean (csrc/spmm.cpp:88-112 per column block):
               grad of buffer c   = A[:, block c]^T g      partial product over the block's CSC arrays (built once)
               -> reduce-scatter to the owners, enqueued as soon as the block's product is queued: it runs under the
               next block's product; the local block comes last and needs no collective;
               grad_value[block]  = SDDMM of the block (tsamd_spmm_value_bw) against the buffer it multiplied.
    backward, min / max (csrc/spmm.cpp:204-242 per column block): the winner ids of the whole result are translated to
               each block's entry numbering (entries of other blocks read as "no winner") and the block's scatter
               kernel (tsamd_spmm_minmax_bw) yields grad_value[block] and the gradient of the block's buffer, which
               is reduce-scattered like above -- the winner's gradient reaches the rank that owns the winning row of X.
    Every rank issues the same collectives whatever its own inputs require (the differentiable path is entered by
    all ranks or by none)."""

    @staticmethod
    def forward(ctx, x_local: Tensor, value: Optional[Tensor], plan, reduce: str):
        x_pad = plan.wire_order(x_local.detach())
        bufs = plan._start_gathers(x_pad)
        works, plan._works = plan._works, []
        vals = None if value is None else [value.detach()[st['src']] for st in plan.stages]
        out, arg = plan.multiply_landed(x_pad, bufs, reduce, works, stage_values=vals)
        ctx.plan, ctx.reduce, ctx.has_value = plan, reduce, value is not None
        ctx.n_local = x_local.size(0)
        ctx.mats = [x_pad] + list(bufs)
        ctx.mat_versions = [m._version for m in ctx.mats]  # (plain attributes, not save_for_backward: checked by hand)
        ctx.vals = vals
        ctx.arg = arg
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        plan, reduce = ctx.plan, ctx.reduce
        g = g.contiguous()
        mats, vals, arg = ctx.mats, ctx.vals, ctx.arg
        if mats is None:
            # the landed buffers (up to the whole of X) are freed by the first backward: keeping them for
            # retain_graph=True would hold N x F elements per step
            raise RuntimeError('_OverlappedProduct: backward called a second time; the gathered operand of the forward was '
                               'released by the first one (retain_graph=True is not supported by the overlapped all-gather; '
                               'use RowShardedSpMM for that)')
        if any(m._version != ver for m, ver in zip(mats, ctx.mat_versions)):
            raise RuntimeError('_OverlappedProduct: a buffer saved for the backward was modified in place after the forward')
        need_v = ctx.has_value and ctx.needs_input_grad[1]
        minmax = reduce in ('min', 'max')
        if reduce == 'mean':
            deg = (plan.rowptr[1:] - plan.rowptr[:-1]).clamp(min=1).to(g.dtype)
            g = g / deg.view(-1, *([1] * (g.dim() - 1)))
        gv = None
        if need_v:
            gv = torch.zeros(plan.E, dtype=mats[0].dtype, device=g.device)
        extra = plan.training_arrays()
        arg_c, stage_of_arg = None, None
        if minmax:
            none = arg >= plan.E
            arg_c = arg.masked_fill(none, 0)
            stage_of_arg = extra['stage_of'][arg_c].masked_fill(none, -1)
            arg_c = extra['local_id'][arg_c]
        pieces, works = [None] * plan.chunks, []
        g0 = None
        for i in list(range(1, len(plan.stages))) + [0]:  # remote blocks first: their collectives run under what follows
            st, ex = plan.stages[i], extra['stages'][i]
            Ei = st['col'].numel()
            v_i = None if vals is None else vals[i]
            if Ei == 0:  # a column block without entries on this rank (unequal shards): its buffer's gradient is zero,
                gval_i, gmat_i = None, torch.zeros_like(mats[i])  # but the collective still has to be joined
            elif minmax:
                arg_i = torch.where(stage_of_arg == i, arg_c, torch.full_like(arg_c, Ei))
                gval_i, gmat_i = plan.minmax_bw_fn(st['rowptr'], st['col'], v_i, mats[i], g, arg_i, need_v)
            else:
                gmat_i = mats[i].new_empty(mats[i].shape)
                vt = None if v_i is None else v_i[ex['perm']]
                plan.partial_fn(ex['colptr'], ex['rowidx'], vt, g, 'sum', gmat_i, None, None, Ei, False, None)
                gval_i = plan.value_bw_fn(ex['row'], st['rowptr'], st['col'], mats[i], g) if need_v else None
            if need_v and gval_i is not None:
                gv[st['src']] = gval_i.to(gv.dtype)
            if i == 0:
                g0 = gmat_i
            else:
                o, w = plan._reduce_scatter(gmat_i)
                pieces[i - 1] = o
                works.append(w)
        for w in works:
            if w is not None:
                w.wait()
        grad_pad = g0 + torch.cat(pieces, 0)
        grad_x = grad_pad.index_select(0, plan.positions[plan.rank][:ctx.n_local])  # row i sits at position positions[i]
        ctx.mats = ctx.vals = ctx.arg = None
        return grad_x, gv, None, None


class OverlappedAllGatherSpMM(object):
    """``RowShardedSpMM`` with the all-gather hidden behind the compute (BASELINE.json north_star:
    "RCCL all-gather of the dense features over xGMI before SpMM").

    Setup (once per matrix): the local row block is split into ``chunks + 1`` column blocks
    (``build_column_stages``) and every rank agrees on the row order of the shards on the wire -- the hashed
    positions of the relabelled layout, so that what lands is ALREADY free of channel camping and the gathered
    operand (17 GB at configs[4]) is never copied again (the serial path pays that copy on every call).
    A step:
        x_h = x_local in wire order (one pass over the LOCAL shard only)
        enqueue all_gather(chunk c of x_h) for c = 0 .. chunks-1      RCCL stream, back to back, full mesh each
        out  = A[:, own columns] @ x_h                                 needs no exchange: runs under collective 0
        out += A[:, chunk c of the other owners] @ buffer c            as soon as collective c has completed
    On a full xGMI mesh every collective uses all links (each peer's chunk arrives over its own link), which a
    per-owner schedule of point-to-point steps would not; the column blocks are "by owner AND chunk" instead.
    sum / mean / min / max; out (and arg_out) equal the serial product -- exactly for min / max (ties between
    blocks go to the smaller entry id), up to the association of the partial sums for sum / mean.
    Differentiable for sum / mean / min / max: the training step is one autograd node (`_OverlappedProduct`) around the
    same kernels; its backward multiplies every column block's transpose (or scatters the winners' gradients, min /
    max) and reduce-scatters each buffer's gradient to the owners under the next block's product.

    ``partial_fn(rowptr, col, value, mat, reduce, out, arg_out, arg_map, arg_none, accumulate, deg_rowptr)``,
    ``value_bw_fn(row, rowptr, col, mat, grad)`` and ``minmax_bw_fn(rowptr, col, value, mat, grad, arg, want_value)``
    are injectable for the CPU (gloo) tests; the product defaults are tsamd_spmm_partial / tsamd_spmm_value_bw /
    tsamd_spmm_minmax_bw."""

    def __init__(self, rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                 group=None, spmm_fn: Optional[Callable] = None, chunks: int = 4,
                 partial_fn: Optional[Callable] = None, positions_fn: Optional[Callable] = None,
                 value_bw_fn: Optional[Callable] = None, minmax_bw_fn: Optional[Callable] = None,
                 agree: str = 'never'):
        assert agree in ('never', 'always', 'once')
        # how the ranks settle the autograd path of a call whose `differentiable` is left to be decided (the backward
        # issues world-wide reduce-scatters: ranks that disagree deadlock).  'never' (default): every rank decides from
        # its own grad mode / requires_grad -- right when all ranks run the same training loop; 'always' / 'once': one
        # all_reduce per call / per local state first, as PipelinedHaloSpMM does
        self.agree = agree
        self.agreements = 0
        self._agreed = {}
        self.group = group
        self.spmm_fn = spmm_fn or _default_spmm
        self.partial_fn = partial_fn or _native_partial
        self.value_bw_fn = value_bw_fn or _native_value_bw
        self.minmax_bw_fn = minmax_bw_fn or _native_minmax_bw
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(x_sizes) == self.world
        self.x_sizes = list(x_sizes)
        self.rowptr, self.col, self.value = rowptr, col, value
        self.rows = rowptr.numel() - 1
        self.E = col.numel()
        self.chunks = max(1, int(chunks))
        self._works = []
        if self.world == 1:
            return
        positions_fn = positions_fn or _default_positions
        dev = col.device
        self.positions = [positions_fn(int(n), dev) for n in self.x_sizes]  # the same on every rank by construction
        self.cs, self.stages = build_column_stages(rowptr, col, self.x_sizes, self.rank, self.chunks, self.positions)
        # wire order of the local shard: x_pad[j] = x_local[inv[j]] for j < n_local (rows behind it are padding)
        mine = self.positions[self.rank]
        inv = torch.empty_like(mine)
        inv[mine] = torch.arange(mine.numel(), dtype=torch.long, device=dev)
        pad = self.chunks * self.cs - mine.numel()
        self.inv = torch.cat([inv, inv.new_zeros(pad)]) if pad > 0 else inv

    # ---- exchange ----------------------------------------------------------------------------
    def wire_order(self, x_local: Tensor) -> Tensor:
        """[chunks * cs, K]: the local shard in the row order it is stored and sent in (padding rows are never read)."""
        if x_local.size(0) == 0:  # a rank that owns no row of X still takes part in every collective
            return x_local.new_zeros((self.inv.numel(), ) + tuple(x_local.shape[1:]))
        return pack_rows(x_local, self.inv) if not x_local.requires_grad else x_local.index_select(0, self.inv)

    def _start_gathers(self, x_pad: Tensor, reuse: bool = False):
        """Enqueue the chunk collectives.  reuse=True (the inference step): the landing buffers of the previous step are
        written again instead of allocating world * rows * K bytes per step (17 GB at configs[4], N = 8) -- a host that
        runs a few steps ahead of the device would otherwise hold several sets, since buffers a collective stream still
        owns cannot be handed out again.  Safe: a collective is ordered behind everything the compute stream was given
        before it, i.e. behind the previous step's products that read these buffers."""
        cs = self.cs
        shape = (self.world * cs, ) + tuple(x_pad.shape[1:])
        cached = getattr(self, '_landing', None)
        if reuse and cached is not None and len(cached) == self.chunks and all(
                b.shape == shape and b.dtype == x_pad.dtype and b.device == x_pad.device for b in cached):
            bufs = cached
        else:
            bufs = [x_pad.new_empty(shape) for _ in range(self.chunks)]
            if reuse:
                self._landing = bufs
        self._works = []
        for c in range(self.chunks):
            self._works.append(dist.all_gather_into_tensor(bufs[c], x_pad[c * cs:(c + 1) * cs].detach(), group=self.group,
                                                           async_op=True))
        return list(bufs)

    def gather_all(self, x_local: Tensor):
        """The exchange alone (for timing): all chunk collectives, waited for."""
        bufs = self._start_gathers(self.wire_order(x_local))
        for w in self._works:
            w.wait()
        self._works = []
        return bufs

    # ---- the product -------------------------------------------------------------------------
    def _stage_value(self, st):
        return None if self.value is None else self.value.detach()[st['src']]

    def multiply_landed(self, x_pad: Tensor, bufs, reduce: str, works=None, stage_values=None):
        """The staged product on buffers that have landed (or land as `works` complete) -> (out, arg_out or None).
        `stage_values`: the per-stage weights of THIS call (the training step); None = the cached copies of self.value."""
        minmax = reduce in ('min', 'max')
        out = x_pad.new_empty((self.rows, ) + tuple(x_pad.shape[1:]))
        arg = torch.empty(out.shape, dtype=torch.long, device=out.device) if minmax else None
        last = len(self.stages) - 1
        # the per-stage copies of the edge weights are gathered once per STATE of the value tensor: identity, storage
        # and version counter (an optimizer step updates a trainable value in place: the next inference call must
        # see it)
        vkey = None if self.value is None else (id(self.value), self.value.data_ptr(), self.value._version)
        if stage_values is None and getattr(self, '_stage_values_key', 0) != vkey:
            self._stage_values = [self._stage_value(st) for st in self.stages]
            self._stage_values_key = vkey
        svals = stage_values if stage_values is not None else self._stage_values
        for i, st in enumerate(self.stages):
            mat = x_pad if i == 0 else bufs[i - 1]
            if i > 0 and works is not None:
                works[i - 1].wait()  # the compute stream waits for collective i - 1, the host does not
            red = reduce
            if reduce == 'mean' and i < last:
                red = 'sum'
            self.partial_fn(st['rowptr'], st['col'], svals[i], mat, red, out, arg,
                            st['src'] if minmax else None, self.E, i > 0, self.rowptr if red == 'mean' else None)
        return out, arg

    def __call__(self, x_local: Tensor, reduce: str = 'sum', differentiable: Optional[bool] = None,
                 return_arg: bool = False):
        """`differentiable`: record for autograd (its backward issues collectives, so EVERY rank must take the same
        path; None = decide from the local grad mode / requires_grad -- pass it explicitly when ranks may differ)."""
        if reduce == 'add':
            reduce = 'sum'
        if self.world == 1:
            return self.spmm_fn(self.rowptr, self.col, self.value, x_local, reduce)
        if differentiable is None:
            differentiable = torch.is_grad_enabled() and (
                x_local.requires_grad or (self.value is not None and self.value.requires_grad))
            if self.agree != 'never':
                key = (torch.is_grad_enabled(), differentiable)
                verdict = self._agreed.get(key) if self.agree == 'once' else None
                if verdict is None:
                    flag = torch.tensor([1 if differentiable else 0], dtype=torch.int32, device=x_local.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                    self.agreements += 1
                    verdict = bool(int(flag))
                    if self.agree == 'once':
                        self._agreed[key] = verdict
                differentiable = verdict
        if differentiable:
            return self._differentiable(x_local, reduce)
        x_pad = self.wire_order(x_local.detach())
        bufs = self._start_gathers(x_pad, reuse=True)
        works, self._works = self._works, []
        out, arg = self.multiply_landed(x_pad, bufs, reduce, works)
        return (out, arg) if return_arg else out

    def _differentiable(self, x_local: Tensor, reduce: str) -> Tensor:
        """The training step: one autograd node (`_OverlappedProduct`) around the inference step's kernels."""
        return _OverlappedProduct.apply(x_local, self.value, self, reduce)

    def training_arrays(self):
        """What only the backward needs, built on first use: per column block the CSC view (colptr over the rows of the
        buffer it multiplies, the local row of every entry in column order, the CSR -> CSC permutation) and the local
        row of every entry in CSR order; for min / max the block and the block-local id of every entry of the row
        block.  Index arrays only: the weights are gathered per step."""
        cached = getattr(self, '_training', None)
        if cached is not None:
            return cached
        dev = self.col.device
        M = self.rows
        stage_of = torch.empty(self.E, dtype=torch.long, device=dev)
        local_id = torch.empty(self.E, dtype=torch.long, device=dev)
        stages = []
        for i, st in enumerate(self.stages):
            src, col = st['src'], st['col']
            n_i = self.chunks * self.cs if i == 0 else self.world * self.cs
            row = torch.repeat_interleave(torch.arange(M, dtype=torch.long, device=dev), st['rowptr'][1:] - st['rowptr'][:-1])
            perm = torch.sort(col, stable=True)[1]
            colptr = torch.zeros(n_i + 1, dtype=torch.long, device=dev)
            if col.numel() > 0:
                torch.cumsum(torch.bincount(col, minlength=n_i), 0, out=colptr[1:])
            stages.append(dict(row=row, perm=perm, colptr=colptr, rowidx=row[perm].contiguous()))
            stage_of[src] = i
            local_id[src] = torch.arange(src.numel(), dtype=torch.long, device=dev)
        self._training = dict(stages=stages, stage_of=stage_of, local_id=local_id)
        return self._training

    def _reduce_scatter(self, gbuf: Tensor):
        """Gradient of a landed buffer [world * cs, K] -> this rank's chunk [cs, K], asynchronously -> (chunk, work)."""
        cs = self.cs
        gbuf = gbuf.contiguous()
        if dist.get_backend(self.group) != 'gloo':
            o = gbuf.new_empty((cs, ) + tuple(gbuf.shape[1:]))
            return o, dist.reduce_scatter_tensor(o, gbuf, group=self.group, async_op=True)
        dist.all_reduce(gbuf, group=self.group)  # gloo has no reduce_scatter
        return gbuf[self.rank * cs:(self.rank + 1) * cs].clone(), None


class _ExchangeRows(torch.autograd.Function):
    """all_to_all of the feature rows each rank asked for; backward routes the gradient of every
    received row back to its owner and adds it up there."""

    @staticmethod
    def forward(ctx, x_local: Tensor, serve_idx: Tensor, send_counts, recv_counts, group):
        ctx.group, ctx.send_counts, ctx.recv_counts = group, list(send_counts), list(recv_counts)
        ctx.save_for_backward(serve_idx)
        ctx.n_local = x_local.size(0)
        send = pack_rows(x_local.detach(), serve_idx)  # rows packed by requesting rank
        recv = x_local.new_empty((sum(recv_counts), ) + tuple(x_local.shape[1:]))
        dist.all_to_all_single(recv, send, output_split_sizes=ctx.recv_counts,
                               input_split_sizes=ctx.send_counts, group=group)
        return recv

    @staticmethod
    def backward(ctx, grad_recv: Tensor):
        (serve_idx, ) = ctx.saved_tensors
        grad_recv = grad_recv.contiguous()
        back = grad_recv.new_empty((sum(ctx.send_counts), ) + tuple(grad_recv.shape[1:]))
        dist.all_to_all_single(back, grad_recv, output_split_sizes=ctx.send_counts,
                               input_split_sizes=ctx.recv_counts, group=ctx.group)
        grad_local = grad_recv.new_zeros((ctx.n_local, ) + tuple(grad_recv.shape[1:]))
        grad_local.index_add_(0, serve_idx, back)
        return grad_local, None, None, None, None


class HaloShardedSpMM(object):
    """Row-sharded SpMM that moves only the rows of X a rank's block actually references.

    An all-gather ships (P-1)/P of X to every rank although a row block with E_local entries
    touches at most E_local (on power-law graphs far fewer) distinct columns; its cost grows with
    the number of ranks while the compute per rank stays constant.  Here the set of referenced
    columns is computed once per matrix (``unique(col)``), each owner learns which of its rows every
    other rank needs (one index all_to_all at setup), and a step is

        packed = X_local[serve_idx]                  (gather kernel)
        X_need = all_to_all_single(packed)           (RCCL, uneven splits, the only collective)
        out_local = A_local' @ X_need                (A_local' = A_local with compacted column ids)

    Same result as ``RowShardedSpMM``: the compaction is a relabelling of columns.
    """

    def __init__(self, rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                 group=None, spmm_fn: Optional[Callable] = None):
        self.group = group
        self.spmm_fn = spmm_fn or _default_spmm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(x_sizes) == self.world
        self.rowptr, self.value = rowptr, value
        self.x_sizes = list(x_sizes)
        if self.world == 1:
            self.col = col
            return
        dev = col.device
        needed = torch.unique(col)  # sorted global column ids this block touches
        bounds = torch.tensor([0] + list(torch.tensor(self.x_sizes).cumsum(0).tolist()), device=dev)
        cuts = torch.searchsorted(needed, bounds)  # needed[cuts[r]:cuts[r+1]] live on rank r
        self.recv_counts = (cuts[1:] - cuts[:-1]).tolist()
        # The order of the rows inside each owner's chunk of X_need is ours to choose, and compact
        # column ids are just positions in X_need.  A seeded random order per chunk scatters the hub
        # columns of Kronecker-like graphs over the memory channels for free (what tsamd_spmm's
        # "relabel" copy buys on a single GPU) -- results do not depend on it.
        g = torch.Generator(device=dev)
        g.manual_seed(0x5eed + self.rank)
        owner = torch.repeat_interleave(torch.arange(self.world, device=dev),
                                        torch.tensor(self.recv_counts, device=dev))
        order = torch.argsort(owner.double() + torch.rand(needed.numel(), generator=g, device=dev).double())
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=dev)
        sorted_needed = needed
        needed = needed[order]  # chunked by owner, shuffled inside each chunk
        # tell every owner how many rows we want, then which ones (owner-local row ids)
        want = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
        serve = torch.empty_like(want)
        dist.all_to_all_single(serve, want, group=group)
        self.send_counts = serve.tolist()
        owner_base = torch.repeat_interleave(bounds[:-1], want)
        req_local = needed - owner_base
        self.serve_idx = torch.empty(sum(self.send_counts), dtype=torch.int64, device=dev)
        dist.all_to_all_single(self.serve_idx, req_local, output_split_sizes=self.send_counts,
                               input_split_sizes=self.recv_counts, group=group)
        self.col = inv[torch.searchsorted(sorted_needed, col)]  # compact ids = positions in X_need
        self.n_needed = needed.numel()

    def exchange(self, x_local: Tensor) -> Tensor:
        if self.world == 1:
            return x_local
        return _ExchangeRows.apply(x_local, self.serve_idx, self.send_counts, self.recv_counts,
                                   self.group)

    def __call__(self, x_local: Tensor, reduce: str = 'sum') -> Tensor:
        return self.spmm_fn(self.rowptr, self.col, self.value, self.exchange(x_local), reduce)


class PipelinedHaloSpMM(object):
    """HaloShardedSpMM with the exchange hidden behind the compute (SUM / MEAN / MIN / MAX all work
    because rows are never split; differentiable w.r.t. X and the values -- the training path queues
    the same exchanges through ``_PipelinedFetch``).

    The local row block is cut into `chunks` pieces of equal nnz.  Piece i needs the column set
    S_i; what has to arrive before it can run is only D_i = S_i minus (S_0 u ... u S_{i-1}) -- hub columns are
    fetched once, with the first piece.  A step is a software pipeline

        exchange(D_0); for i: [exchange(D_{i+1}) on the RCCL stream]  ||  [SpMM(piece i)]

    so only the first exchange is exposed.  Total traffic equals the one-shot halo exchange.
    X_need is one buffer laid out [D_0 | D_1 | ...]; piece i multiplies against its prefix.
    """

    def __init__(self, rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                 group=None, spmm_fn: Optional[Callable] = None, chunks: int = 4,
                 spmm_into_fn: Optional[Callable] = None, agree: str = 'always'):
        assert agree in ('always', 'once')
        self.agree = agree      # how the ranks settle the autograd path of a call (see __call__)
        self.agreements = 0     # collectives spent on it so far
        self.group = group
        # with the product kernels the pieces write into one preallocated result (no torch.cat); an
        # injected spmm_fn (CPU tests) gets the concatenating path unless it brings its own writer
        self.spmm_into_fn = spmm_into_fn if spmm_into_fn is not None else (
            _default_spmm_into if spmm_fn is None else None)
        self._works, self._keep = [], []
        self._agreed = {}  # local (grad mode, needs grad) -> the path every rank agreed on (see __call__)
        self.spmm_fn = spmm_fn or _default_spmm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(x_sizes) == self.world
        self.x_sizes = list(x_sizes)
        dev = col.device
        M = rowptr.numel() - 1
        self.pieces = []
        ranges = partition_rows(rowptr, chunks, 'nnz') if self.world > 1 else [(0, M)]
        bounds = torch.tensor([0] + list(torch.tensor(list(x_sizes)).cumsum(0).tolist()), device=dev)
        have = torch.zeros(0, dtype=torch.int64, device=dev)  # sorted global ids already in the buffer
        have_pos = torch.zeros(0, dtype=torch.int64, device=dev)  # their positions in the buffer
        offset = 0
        for (s, e) in ranges:
            rp, c, v = narrow_rows(rowptr, col, value, s, e)
            if self.world == 1:
                self.pieces.append(dict(rows=(s, e), rowptr=rp, col=c, value=v, n_prefix=int(x_sizes[0])))
                continue
            need = torch.unique(c)
            if have.numel() > 0:
                idx = torch.searchsorted(have, need).clamp_(max=have.numel() - 1)
                fresh = need[have[idx] != need]
            else:
                fresh = need
            cuts = torch.searchsorted(fresh, bounds)
            recv_counts = (cuts[1:] - cuts[:-1]).tolist()
            want = torch.tensor(recv_counts, dtype=torch.int64, device=dev)
            serve = torch.empty_like(want)
            dist.all_to_all_single(serve, want, group=group)
            send_counts = serve.tolist()
            req_local = fresh - torch.repeat_interleave(bounds[:-1], want)
            serve_idx = torch.empty(sum(send_counts), dtype=torch.int64, device=dev)
            dist.all_to_all_single(serve_idx, req_local, output_split_sizes=send_counts,
                                   input_split_sizes=recv_counts, group=group)
            # merge the new ids into the (sorted ids -> buffer position) map
            pos_new = offset + torch.arange(fresh.numel(), device=dev)
            allid = torch.cat([have, fresh])
            allpos = torch.cat([have_pos, pos_new])
            order = torch.argsort(allid)
            have, have_pos = allid[order], allpos[order]
            offset += fresh.numel()
            ccompact = have_pos[torch.searchsorted(have, c)]
            self.pieces.append(dict(rows=(s, e), rowptr=rp, col=ccompact, value=v, serve_idx=serve_idx,
                                    send_counts=send_counts, recv_counts=recv_counts,
                                    seg=(offset - fresh.numel(), offset), n_prefix=offset))
        self.n_needed = offset if self.world > 1 else int(x_sizes[0])
        self.rows = M

    def _start_exchange(self, piece, x_local: Tensor, buf: Tensor):
        send = pack_rows(x_local.detach(), piece['serve_idx'])
        a, b = piece['seg']
        return dist.all_to_all_single(buf[a:b], send, output_split_sizes=piece['recv_counts'],
                                      input_split_sizes=piece['send_counts'], group=self.group,
                                      async_op=True), send

    def __call__(self, x_local: Tensor, reduce: str = 'sum', differentiable: Optional[bool] = None) -> Tensor:
        """`differentiable`: take the autograd-recording path (its backward issues collectives, so EVERY rank
        must take the same path).  None (default) = any rank needs a gradient: the local verdict (grad mode,
        requires_grad of X / the values) is agreed on with one all_reduce, so that a rank whose inputs happen not
        to require grad cannot leave its peers waiting in a backward collective.  ``agree='always'`` (default)
        does that on EVERY call -- correct for any program, at the price of a collective and a read-back in front
        of the pipelined exchange.  ``agree='once'`` does it the first time the plan sees a local state and
        remembers the verdict per state: only for SPMD programs whose ranks change grad mode / requires_grad in
        lockstep (a rank that meets a new state alone would pair its all_reduce with its peers' next all_to_all).
        Loops that know their path pass True / False and pay nothing (bench.py does)."""
        if self.world == 1:
            p = self.pieces[0]
            return self.spmm_fn(p['rowptr'], p['col'], p['value'], x_local, reduce)
        if differentiable is None:
            needs_grad = torch.is_grad_enabled() and (
                x_local.requires_grad or any(p['value'] is not None and p['value'].requires_grad for p in self.pieces))
            # agree == 'once': the agreement (a collective + a read-back, which serialise the pipelined exchange) is
            # made once per local state and remembered
            key = (torch.is_grad_enabled(), needs_grad)
            verdict = self._agreed.get(key) if self.agree == 'once' else None
            if verdict is None:
                flag = torch.tensor([1 if needs_grad else 0], dtype=torch.int32, device=x_local.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                self.agreements += 1
                verdict = bool(int(flag))
                if self.agree == 'once':
                    self._agreed[key] = verdict
            differentiable = verdict
        if differentiable:
            return self._differentiable(x_local, reduce)
        buf = x_local.new_empty((self.n_needed, ) + tuple(x_local.shape[1:]))
        # one result buffer for the whole row block: every piece writes its rows in place
        out = None
        if self.spmm_into_fn is not None:
            out = x_local.new_empty((self.rows, ) + tuple(x_local.shape[1:]))
        outs = []
        work, keep = self._start_exchange(self.pieces[0], x_local, buf)
        for i, p in enumerate(self.pieces):
            nxt = None
            if i + 1 < len(self.pieces):  # enqueue the next exchange BEFORE this piece's SpMM
                nxt = self._start_exchange(self.pieces[i + 1], x_local, buf)
            work.wait()  # this piece's rows have landed (the compute stream waits, the host does not)
            if out is not None:
                s, e = p['rows']
                self.spmm_into_fn(p['rowptr'], p['col'], p['value'], buf[:p['n_prefix']], reduce, out[s:e])
            else:
                outs.append(self.spmm_fn(p['rowptr'], p['col'], p['value'], buf[:p['n_prefix']], reduce))
            if nxt is not None:
                work, keep = nxt
        return out if out is not None else torch.cat(outs, dim=-2)

    def _differentiable(self, x_local: Tensor, reduce: str) -> Tensor:
        """Training step: the same exchanges, recorded for autograd.  All pieces' exchanges are queued up
        front (RCCL runs them in order while the pieces multiply); the gradient of the fetched rows is
        summed over the pieces by autograd and routed back to the owners piece by piece."""
        buf = _PipelinedFetch.apply(x_local, self)
        outs = []
        for i, p in enumerate(self.pieces):
            self._works[i].wait()
            outs.append(self.spmm_fn(p['rowptr'], p['col'], p['value'], buf[:p['n_prefix']], reduce))
        self._works = []
        return torch.cat(outs, dim=-2)


class _PipelinedFetch(torch.autograd.Function):
    """X_need = [D_0 | D_1 | ...] of a PipelinedHaloSpMM with every piece's all_to_all enqueued
    asynchronously; backward sends each segment's gradient back to the owners and adds it up there."""

    @staticmethod
    def forward(ctx, x_local: Tensor, plan):
        ctx.plan = plan
        ctx.n_local = x_local.size(0)
        buf = x_local.new_empty((plan.n_needed, ) + tuple(x_local.shape[1:]))
        plan._works, plan._keep = [], []
        for p in plan.pieces:
            work, send = plan._start_exchange(p, x_local, buf)
            plan._works.append(work)
            plan._keep.append(send)  # the send buffers must outlive the collectives
        return buf

    @staticmethod
    def backward(ctx, grad_buf: Tensor):
        plan = ctx.plan
        grad_buf = grad_buf.contiguous()
        grad_local = grad_buf.new_zeros((ctx.n_local, ) + tuple(grad_buf.shape[1:]))
        for p in plan.pieces:
            a, b = p['seg']
            back = grad_buf.new_empty((sum(p['send_counts']), ) + tuple(grad_buf.shape[1:]))
            dist.all_to_all_single(back, grad_buf[a:b].contiguous(), output_split_sizes=p['send_counts'],
                                   input_split_sizes=p['recv_counts'], group=plan.group)
            grad_local.index_add_(0, p['serve_idx'], back)
        plan._keep = []
        return grad_local, None


def shard_matrix(rowptr: Tensor, col: Tensor, value: Optional[Tensor], n_cols: int, group=None,
                 balance: str = 'nnz', spmm_fn: Optional[Callable] = None, exchange: str = 'allgather', **kw):
    """Convenience for a replicated global CSR: every rank cuts out its own row block.  X is
    sharded by the same row ranges when the matrix is square (GNN layers chain that way),
    otherwise in equal blocks of columns."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ranges = partition_rows(rowptr, world, balance)
    M = rowptr.numel() - 1
    if M == n_cols:
        x_sizes = [e - s for s, e in ranges]
    else:
        x_sizes = [(n_cols * (p + 1)) // world - (n_cols * p) // world for p in range(world)]
    s, e = ranges[rank]
    rp, c, v = narrow_rows(rowptr, col, value, s, e)
    cls = EXCHANGES[exchange]
    return cls(rp, c, v, x_sizes, group, spmm_fn, **kw), (s, e)


# ---------------------------------------------------------------------------------------------
# orchestration helpers of the multi-GPU benchmark (bench.py); backend agnostic so that the control
# flow is exercised by the gloo tests (tests/test_parallel_cpu.py)
# ---------------------------------------------------------------------------------------------
EXCHANGES = {'allgather': OverlappedAllGatherSpMM, 'allgather_serial': RowShardedSpMM, 'halo': HaloShardedSpMM,
             'pipelined': PipelinedHaloSpMM}


def build_with_fallback(rowptr: Tensor, col: Tensor, value: Optional[Tensor], x_sizes: Sequence[int],
                        x_local: Tensor, reduce: str, spmm_fn: Callable, requested: str, chunks: int = 8,
                        group=None, sync: Optional[Callable] = None, ag_chunks: int = 4,
                        partial_fn: Optional[Callable] = None):
    """Plan the requested exchange and run one trial step; if that raises a RuntimeError (an unsupported
    collective, an allocation that does not fit ... -- the exception class goes into the reason; other
    exception types are bugs and propagate) the ranks agree through an all_reduce and fall back together:
    requested -> halo -> allgather (overlapped) -> allgather_serial.  Meant for failures every rank hits at the same point; a rank that
    dies in the middle of a collective sequence leaves its peers waiting, nothing recovers from that.
    -> (sharded operator, mode actually used, None or the reason of the first fall-back)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    sync = sync or (lambda: None)
    reason = None
    for mode in [requested] + [m for m in ('halo', 'allgather', 'allgather_serial') if m != requested]:
        err = None
        try:
            cls = EXCHANGES[mode] if world > 1 else RowShardedSpMM
            kw = dict(chunks=chunks) if cls is PipelinedHaloSpMM else {}
            if cls is OverlappedAllGatherSpMM:
                kw = dict(chunks=ag_chunks, partial_fn=partial_fn)
            sharded = cls(rowptr, col, value, x_sizes, group, spmm_fn, **kw)
            with torch.no_grad():
                if cls in (PipelinedHaloSpMM, OverlappedAllGatherSpMM):
                    sharded(x_local, reduce, differentiable=False)
                else:
                    sharded(x_local, reduce)
            sync()
        except RuntimeError as exc:  # incl. torch.cuda.OutOfMemoryError and the c10 / RCCL errors; anything else
            err = '%s: %s' % (type(exc).__name__, str(exc)[:200])  # (TypeError, AssertionError: a bug) propagates
        failed = torch.tensor([0 if err is None else 1], device=x_local.device)
        if world > 1:
            dist.all_reduce(failed, op=dist.ReduceOp.MAX, group=group)
        if int(failed) == 0:
            return sharded, mode, reason
        reason = reason or ('%s failed (%s)' % (mode, err or 'on another rank'))
        if world == 1:
            raise RuntimeError(err)
    raise RuntimeError('no exchange mode works: %s' % reason)


def exchange_breakdown(sharded, ref_plan, x_local: Tensor, spmm_call: Callable, n_global: int,
                       row_bytes: int, reps: int = 5, group=None, sync: Optional[Callable] = None,
                       link_gbs: float = 153.0) -> dict:
    """N > 1 only: time the exchange alone (one-shot fetch of the rows this rank's block references)
    and the local SpMM alone (`spmm_call()`), max over ranks, and put the xGMI model of the exchange
    next to it: every peer's share arrives over its own link at `link_gbs` GB/s peak."""
    import time
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sync = sync or (lambda: None)
    if isinstance(sharded, RowShardedSpMM):
        fetch = lambda: sharded.gather(x_local)  # noqa: E731
        rows_in = n_global - x_local.size(0)
        mode = 'allgather_serial'
    elif isinstance(sharded, OverlappedAllGatherSpMM):
        fetch = lambda: sharded.gather_all(x_local)  # noqa: E731
        rows_in = n_global - x_local.size(0)
        mode = 'allgather'
    else:
        plan = sharded if isinstance(sharded, HaloShardedSpMM) else ref_plan
        fetch = lambda: plan.exchange(x_local)  # noqa: E731
        rows_in = int(plan.n_needed - plan.recv_counts[rank])  # rows that cross a link
        mode = 'halo' if isinstance(sharded, HaloShardedSpMM) else 'pipelined'
    t_parts = []
    for fn in (fetch, spmm_call):
        with torch.no_grad():
            fn()
        sync()
        dist.barrier(group)
        t1 = time.perf_counter()
        with torch.no_grad():
            for _ in range(reps):
                fn()
                # one call in flight at a time: an all-gather allocates its landing buffers per call (17 GB at configs[4],
                # N = 8), and buffers a collective stream still owns cannot be handed out again -- ten calls queued
                # back to back would hold ten sets
                sync()
        dist.barrier(group)
        t_parts.append((time.perf_counter() - t1) / reps * 1e3)
    part = torch.tensor(t_parts + [float(rows_in)], dtype=torch.float64, device=x_local.device)
    dist.all_reduce(part, op=dist.ReduceOp.MAX, group=group)
    bytes_in = float(part[2]) * row_bytes
    return dict(mode=mode, exchange_only_ms=round(float(part[0]), 3), spmm_only_ms=round(float(part[1]), 3),
                max_rows_in_per_rank=int(part[2]), max_bytes_in_per_rank=int(bytes_in),
                modelled_exchange_ms=round(bytes_in / (world - 1) / (link_gbs * 1e9) * 1e3, 3),
                model='bytes_in / (N - 1) peers, each over its own xGMI link at %.0f GB/s peak' % link_gbs)

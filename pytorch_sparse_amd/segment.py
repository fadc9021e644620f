"""Differentiable wrapper of ``tsamd::segment_reduce`` (the fused replacement of torch_scatter's
``segment_csr``, which is differentiable in the reference: torch_sparse/reduce.py and
storage.py:457-466 rely on it for ``adj.sum(dim)`` / ``coalesce`` with learnable edge weights).

Forward: the HIP kernel.  Backward (sum / mean: every entry of a segment receives the segment's
gradient, mean divided by the segment length; min / max: only the first entry that attains the
result does, as torch_scatter routes it through ``arg_out``)."""
from typing import Optional

import torch
from torch import Tensor


def _expand(t: Tensor, like: Tensor) -> Tensor:
    return t.view((-1, ) + (1, ) * (like.dim() - 1))


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value: Tensor, perm: Optional[Tensor], seg_ptr: Tensor, nseg: int, reduce: str,
                balanced: bool):
        out = torch.ops.tsamd.segment_reduce(value.detach(), perm, seg_ptr, nseg, reduce, balanced)
        ctx.reduce, ctx.nseg, ctx.balanced = reduce, nseg, balanced
        ctx.has_perm = perm is not None
        saved = [seg_ptr, value if reduce in ('min', 'max') else value.new_empty(0),
                 out if reduce in ('min', 'max') else value.new_empty(0)]
        if perm is not None:
            saved.append(perm)
        ctx.save_for_backward(*saved)
        ctx.n_entries = value.size(0)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        seg_ptr, value, out = ctx.saved_tensors[:3]
        perm = ctx.saved_tensors[3] if ctx.has_perm else None
        E, nseg, reduce = ctx.n_entries, ctx.nseg, ctx.reduce
        ptr = seg_ptr[:nseg + 1]
        seg_id = torch.ops.torch_sparse.ptr2ind(ptr.contiguous(), E)  # segment of every entry
        grad_out = grad_out.contiguous()
        if reduce in ('sum', 'add'):
            g = grad_out.index_select(0, seg_id)
        elif reduce == 'mean':
            cnt = (ptr[1:] - ptr[:-1]).clamp_(min=1).to(grad_out.dtype)
            g = (grad_out / _expand(cnt, grad_out)).index_select(0, seg_id)
        else:
            v = value if perm is None else value.index_select(0, perm)
            hit = v == out.index_select(0, seg_id)
            pos = _expand(torch.arange(E, device=v.device), v).expand_as(v)
            cand = torch.where(hit, pos, torch.full_like(pos, E)).contiguous()
            arg = torch.ops.tsamd.segment_reduce(cand, None, seg_ptr, nseg, 'min', ctx.balanced)  # first hit
            empty = _expand(ptr[1:] == ptr[:-1], arg)
            arg = torch.where(empty, torch.full_like(arg, E), arg)  # empty segments feed nobody
            g = grad_out.new_zeros((E + 1, ) + tuple(grad_out.shape[1:]))
            g.scatter_(0, arg, grad_out)
            g = g[:E]
        if perm is not None:  # entry i of the segmented order is value[perm[i]]
            g = torch.zeros_like(g).index_copy_(0, perm, g)
        return g, None, None, None, None, None


def segment_reduce(value: Tensor, perm: Optional[Tensor], seg_ptr: Tensor, nseg: int, reduce: str,
                   balanced: bool = False) -> Tensor:
    """REDUCE over value[perm][seg_ptr[j]:seg_ptr[j+1]] for j < nseg; differentiable w.r.t. value.
    balanced=True when the segments are matrix rows / columns (possibly hubs): entry-balanced path."""
    if value.requires_grad and torch.is_grad_enabled():
        return _SegmentReduce.apply(value, perm, seg_ptr, nseg, reduce, balanced)
    return torch.ops.tsamd.segment_reduce(value, perm, seg_ptr, nseg, reduce, balanced)

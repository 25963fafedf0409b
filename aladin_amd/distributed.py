"""Caption-block sharding of the alignment triplet loss over the GPUs of one node (one process per
GPU, torch.distributed backend "nccl" == RCCL over xGMI).

The reference is single-device; parity target = the single-device result
on the concatenated global batch.  S[i][j] depends only on image i and caption j, so:

    rank r owns captions [r*B, (r+1)*B) and images [r*B, (r+1)*B)
    1. all-gather the image sets (+ lengths)                       -- the one exchange of the path
    2. S[:, block r] = HIP alignment scores (all images x own captions)
    3. all-gather the (W*B x B) score blocks -> full S on every rank (2 MB per rank at B=256)
    4. hinge on the full S (replicated, identical bits on every rank)
    5. backward: each rank differentiates its own column block; d(image sets) from all ranks are
       summed with a reduce-scatter, d(captions) is local.

`scores_fn` / `hinge_fn` default to the HIP ops; CPU tests (gloo, world_size 2) inject the torch
restatement from oracle/ to exercise the collectives without a GPU.
"""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group), dist.get_rank(group)


class _AllGatherRows(torch.autograd.Function):
    """cat over ranks along dim 0; backward = reduce-scatter(sum) of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        W, _ = _world(group)
        x = x.contiguous()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        W, r = _world(ctx.group)
        g = g.contiguous()
        n = g.shape[0] // W
        if dist.get_backend(ctx.group) == 'gloo':            # gloo has no reduce_scatter
            dist.all_reduce(g, group=ctx.group)
            return g[r * n:(r + 1) * n].clone(), None
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, group=ctx.group)
        return out, None


class _GatherColumnBlocks(torch.autograd.Function):
    """(N x b) block per rank -> (N x W*b) on every rank.  The loss built on the result is
    REPLICATED (same value on every rank), so backward hands each rank its own block of the
    gradient -- no sum over ranks."""

    @staticmethod
    def forward(ctx, blk, group):
        ctx.group = group
        W, _ = _world(group)
        parts = torch.empty((W * blk.shape[0], blk.shape[1]), dtype=blk.dtype, device=blk.device)
        dist.all_gather_into_tensor(parts, blk.contiguous(), group=group)
        return parts.view(W, blk.shape[0], blk.shape[1]).permute(1, 0, 2).reshape(blk.shape[0], W * blk.shape[1])

    @staticmethod
    def backward(ctx, g):
        W, r = _world(ctx.group)
        b = g.shape[1] // W
        return g[:, r * b:(r + 1) * b].contiguous(), None


def gather_lengths(lens, device, group=None):
    W, _ = _world(group)
    t = torch.tensor([int(v) for v in lens], dtype=torch.int32, device=device)
    out = torch.empty(W * t.numel(), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def sharded_alignment_loss(im_set, s_seq, im_len, s_len, margin=0.2, max_violation=True, group=None,
                           scores_fn=None, hinge_fn=None):
    """(loss, S_full) of the alignment triplet loss on the GLOBAL batch (all ranks' samples), every
    rank passing its local (B,R,D) / (B,T,D) sets; all ranks must use the same B, R, T.
    Equals AlignmentContrastiveLoss(...)(cat(im), cat(s), ...) on one device."""
    if scores_fn is None or hinge_fn is None:
        from . import ops
        scores_fn = scores_fn or ops.alignment_scores
        hinge_fn = hinge_fn or ops.hinge_loss
    im_all = _AllGatherRows.apply(im_set, group)
    il_all = gather_lengths(im_len, im_set.device, group)
    S_blk = scores_fn(im_all, s_seq, il_all, s_len)                   # (W*B, B)
    S_full = _GatherColumnBlocks.apply(S_blk, group)
    return hinge_fn(S_full, margin, max_violation), S_full

"""CPU stand-ins for the HIP entry points the fast sharded step calls (aladin_amd/distributed.py: _ShardedTriplet).  TEST INFRASTRUCTURE.

No GPU exists in the CPU tier and no node with more than one GPU has been available to this build, so the node that
`bench.py --gpus N` times is executed here across REAL processes (gloo): install() replaces, in the calling process only, the
ops functions the node calls by torch restatements that keep the C ABI's contracts --

    ops.pack_images / pack_captions   the packed operand LAYOUT of aladin_align_geometry (host-only C call, used as is): fp16
                                      roundings of the unit rows; masked rows zero, tile-filling rows copies of the image's first
                                      scored region (csrc/align_fwd.hip: pack kernels), written into the caller's buffers when given
    ops.scores_from_packed            mask-free: per image the max over its `mrows` main rows (and `rem` side rows), per caption
                                      the sum over its `trows` rows -- from the PACKED operands only, like the score kernels
    ops._hinge_raw                    reference alad/loss.py:42-67 with the first-index rule of torch.max, dS as the kernels emit it
    ops._align_backward               autograd of alad/loss.py:80-125 on the raw fp32 sets for the pairs carrying a gradient
                                      (SURVEY appendix A.4), *gscale applied
    ops._check_sets                   the same checks without the device requirement

-- so every line of _ShardedTriplet, FlatSegments and SparseImageExchange (segment views, rank offsets, the score-block permute,
the exchange choice, the empty-need branch, reduce-scatter / all-to-all) runs unchanged.  Nothing here imports oracle/: the tests
compare the sharded result with the oracle.  The product package never imports this module; `bench.py --cpu-standin` does, to run
its real step loop under gloo, and labels its line accordingly (no performance claim).
"""
import torch
import torch.nn.functional as F

CALLS = {}


def _count(name):
    CALLS[name] = CALLS.get(name, 0) + 1


def _unit_rows(x):
    return F.normalize(x.to(torch.float32), p=2, dim=-1, eps=1e-12)


def pack_images(im, im_len_t, geom, rnorm=None, out=None):
    _count('pack_images')
    Bi, R, D = im.shape
    assert (Bi, R, D) == (geom.Bi, geom.R, geom.D) and not geom.split
    Rq, mrows, rem, Dp = int(geom.Rq), int(geom.mrows), int(geom.rem), int(geom.Dp)
    if out is not None:
        xm, xe = out
    else:
        xm = torch.empty(geom.xm_bytes // 2, dtype=torch.float16)
        xe = torch.empty(max(geom.xe_bytes // 2, 8), dtype=torch.float16)
    xm2 = xm[:int(geom.xm_rows) * Dp].view(int(geom.Bi_pad), mrows, Dp)
    xm2.zero_()
    unit = _unit_rows(im[:, 1:1 + Rq])                                 # region 0 dropped (alad/loss.py:87)
    L = (im_len_t.to(torch.int64) - 1 - int(geom.x_tail)).clamp(0, Rq)     # alad/loss.py:89
    rho = torch.arange(mrows)
    src = torch.where(rho < Rq, rho, torch.zeros_like(rho))            # tile-filling rows repeat the first scored region
    rows = unit[:, src.clamp(max=Rq - 1)]                              # (Bi, mrows, D)
    live = (src[None, :] < L[:, None]).unsqueeze(-1)
    xm2[:Bi, :, :D] = torch.where(live, rows, torch.zeros_like(rows)).to(torch.float16)
    if rem:
        xe2 = xe[:int(geom.xe_rows) * Dp].view(int(geom.xe_rows), Dp)
        xe2.zero_()
        rho_e = mrows + torch.arange(rem)
        rows = unit[:, rho_e]
        live = (rho_e[None, :] < L[:, None]).unsqueeze(-1)
        xe2[:Bi * rem].view(Bi, rem, Dp)[:, :, :D] = torch.where(live, rows, torch.zeros_like(rows)).to(torch.float16)
    return xm, xe


def pack_captions(s, s_len_t, geom, rnorm=None):
    _count('pack_captions')
    Bc, T, D = s.shape
    assert (Bc, T, D) == (geom.Bc, geom.T, geom.D) and not geom.split
    Tq, trows, Dp = int(geom.Tq), int(geom.trows), int(geom.Dp)
    y = torch.zeros(geom.y_bytes // 2, dtype=torch.float16)
    y2 = y.view(int(geom.Bc_pad), trows, Dp)
    unit = _unit_rows(s[:, 1:1 + Tq])                                  # token 0 and the tail dropped (alad/loss.py:88)
    L = (s_len_t.to(torch.int64) - 1 - int(geom.y_tail)).clamp(0, Tq)      # alad/loss.py:90
    live = (torch.arange(Tq)[None, :] < L[:, None]).unsqueeze(-1)
    y2[:Bc, :Tq, :D] = torch.where(live, unit, torch.zeros_like(unit)).to(torch.float16)
    return y


def scores_from_packed(xm, xe, y, geom, out=None, e_scratch=None, reuse_side=False):
    _count('scores_from_packed')
    Bi, Bc, mrows, rem, trows, Dp = (int(v) for v in (geom.Bi, geom.Bc, geom.mrows, geom.rem, geom.trows, geom.Dp))
    S = out if out is not None else torch.empty((Bi, Bc), dtype=torch.float32)
    yf = y[:int(geom.y_rows) * Dp].view(-1, Dp)[:Bc * trows].to(torch.float32)
    xf = xm[:int(geom.xm_rows) * Dp].view(-1, Dp)[:Bi * mrows].to(torch.float32)
    A = (xf @ yf.t()).view(Bi, mrows, Bc, trows).amax(dim=1)                    # max over an image's main rows
    if rem:
        ef = xe[:int(geom.xe_rows) * Dp].view(-1, Dp)[:Bi * rem].to(torch.float32)
        A = torch.maximum(A, (ef @ yf.t()).view(Bi, rem, Bc, trows).amax(dim=1))
    S.copy_(A.sum(dim=2))                                                       # sum over a caption's rows (zero rows add 0)
    return S


def hinge_raw(scores, margin, max_violation, want_grad, want_pairs=False, loss_out=None):
    _count('hinge_raw')
    S = scores.detach().to(torch.float32)
    B = S.shape[0]
    diag = S.diag().view(-1, 1)
    eye = torch.eye(B, dtype=torch.bool)
    cs = (margin + S - diag).clamp(min=0).masked_fill(eye, 0)
    ci = (margin + S - diag.t()).clamp(min=0).masked_fill(eye, 0)
    dS = torch.zeros((B, B), dtype=torch.float32) if want_grad else None
    idx = torch.arange(B)
    if max_violation:
        v_s, j_s = cs.max(1)
        v_i, i_i = ci.max(0)
        loss = v_s.sum() + v_i.sum()
        if want_grad:
            on = (v_s > 0).to(torch.float32)
            dS.index_put_((idx, j_s), on, accumulate=True)
            dS.index_put_((idx, idx), -on, accumulate=True)
            on = (v_i > 0).to(torch.float32)
            dS.index_put_((i_i, idx), on, accumulate=True)
            dS.index_put_((idx, idx), -on, accumulate=True)
    else:
        loss = cs.sum() + ci.sum()
        if want_grad:
            P, Q = (cs > 0).to(torch.float32), (ci > 0).to(torch.float32)
            dS += P + Q - torch.diag(P.sum(1)) - torch.diag(Q.sum(0))
    if loss_out is not None:
        loss_out.copy_(loss)
        loss = loss_out
    pairs = None
    if want_grad and want_pairs:
        nz = dS.reshape(-1).nonzero().reshape(-1).to(torch.int32)
        lst = torch.zeros(B * B, dtype=torch.int32)
        lst[:nz.numel()] = nz
        pairs = (lst, torch.tensor([nz.numel()], dtype=torch.int32))
    return loss.to(torch.float32), dS, pairs


def masked_pair_scores(a, b, La, Lb, Rq, Tq):
    """S[n] of n (image, caption) pairs given as (n, R, D) / (n, T, D) raw sets with their lengths: alad/loss.py:80-125, 'MrSw'."""
    x = _unit_rows(a)[:, 1:1 + Rq]
    w = _unit_rows(b)[:, 1:1 + Tq]
    A = torch.bmm(x, w.transpose(1, 2))                                         # (n, R', T')
    dead = ~((torch.arange(Rq)[None, :, None] < La[:, None, None]) & (torch.arange(Tq)[None, None, :] < Lb[:, None, None]))
    return A.masked_fill(dead, 0.0).amax(dim=1).sum(dim=1)


def align_backward(im, s, im_len_t, s_len_t, dS, gscale=None, packed=None, pairs=None, x_tails=(0, 2), dense=False, fill=None):
    _count('align_backward')
    assert x_tails == (0, 2)
    Bi, R, D = im.shape
    Bc, T, _ = s.shape
    g = dS.to(torch.float32) * (float(gscale) if gscale is not None else 1.0)
    nz = g.nonzero()
    with torch.enable_grad():                                   # called from inside an autograd backward
        a = im.detach().to(torch.float32).clone().requires_grad_(True)
        b = s.detach().to(torch.float32).clone().requires_grad_(True)
        if nz.shape[0]:
            i, j = nz[:, 0], nz[:, 1]
            La = (im_len_t.to(torch.int64) - 1).clamp(0, R - 1)[i]
            Lb = (s_len_t.to(torch.int64) - 3).clamp(0, T - 3)[j]
            Sn = masked_pair_scores(a[i], b[j], La, Lb, R - 1, T - 3)
            (Sn * g[i, j]).sum().backward()
    d_im = a.grad if a.grad is not None else torch.zeros_like(a)
    d_s = b.grad if b.grad is not None else torch.zeros_like(b)
    return d_im, d_s


def check_sets(im_set, s_seq, im_len, s_len):
    if im_set.dim() != 3 or s_seq.dim() != 3:
        raise ValueError('aladin_amd: im_set (B,R,D) and s_seq (B,T,D) expected')
    if len(im_len) != im_set.shape[0] or len(s_len) != s_seq.shape[0]:
        raise ValueError('aladin_amd: one length per sample expected')
    as_t = lambda v: v.to(torch.int32) if isinstance(v, torch.Tensor) else torch.tensor([int(x) for x in v], dtype=torch.int32)
    return as_t(im_len), as_t(s_len)


def pack_sets(im, s, im_len_t, s_len_t, geom, norms=True):
    _count('pack_sets')
    xm, xe = pack_images(im, im_len_t, geom)
    return geom, xm, xe, pack_captions(s, s_len_t, geom), None


def install():
    """Replace the HIP-backed functions of aladin_amd.ops in THIS process (a test worker or a --cpu-standin bench rank)."""
    from aladin_amd import ops
    ops.pack_images = pack_images
    ops.pack_captions = pack_captions
    ops.pack_sets = pack_sets
    ops.scores_from_packed = scores_from_packed
    ops._hinge_raw = hinge_raw
    ops._align_backward = align_backward
    ops._check_sets = check_sets
    CALLS.clear()
    return ops


def single_process_step(im, s, im_len, s_len, margin=0.2, max_violation=True):
    """The unsharded composition of the same stand-ins on one batch: (loss, S, d_im, d_s)."""
    from aladin_amd import ops
    il, sl = check_sets(im, s, im_len, s_len)
    g = ops.align_geometry(im.shape[0], s.shape[0], im.shape[1], s.shape[1], im.shape[2])
    xm, xe = pack_images(im, il, g)
    y = pack_captions(s, sl, g)
    S = scores_from_packed(xm, xe, y, g)
    loss, dS, _ = hinge_raw(S, margin, max_violation, True)
    d_im, d_s = align_backward(im, s, il, sl, dS)
    return loss, S, d_im, d_s

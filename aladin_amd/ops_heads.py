"""The loss heads of a training step as single autograd nodes (reference alad/alad_model.py:371-454 calling alad/loss.py): the
matching hinge, the alignment hinge and ListNet distillation with their weighted sum -- the small-batch kernels at B <= 64 (every
shipped YAML trains with bs 32), the general kernels above."""
import ctypes as C

import torch

from . import _lib
from . import ops as _al
from ._ops_common import _RAW_STREAM, _stream, _ptr, _require_gpu, _rows_inner_contig, _LEN_CACHE, lengths_tensor, _ld, _workspace
from .ops import (_FILL_HINT, _align_backward, _align_forward, _caption_fill, _check_backward_supported, _check_sets, _density_probe,
                  _packed_from_buf, _packed_struct, _pair_kernel_covers, _set_view, _triplet_backward, _triplet_forward)
from .ops_losses import _hinge_raw, _sgemm, dot_scores


# ------------------------------------------------------------------------------------------------
# small-batch matching + distillation (B <= 64: the batch size of every shipped YAML is 32)
# ------------------------------------------------------------------------------------------------
SMALL_BATCH_MAX = 64
HEAD_MATCH_HINGE, HEAD_ALIGN_HINGE, HEAD_LISTNET = 1, 2, 4      # ALADIN_HEAD_* of include/aladin_hip.h


def _heads_small_fwd(im, s, S, margin, max_violation, flags, temperature, eps, weights, want_grads, want_pairs, align=None):
    """Launch aladin_heads_small_fwd -> dict of its outputs (see include/aladin_hip.h).
    align = (im_set, s_seq, im_len_t, s_len_t, packed): with the hardest-negative alignment hinge and the fp16 pair
    kernel's shapes the element-wise pass shares its launch with the backward's pair recompute
    (aladin_heads_small_fwd_argmax); out['table_ws'] then holds the argmax table and out['sets'] the sets in the row
    layout the kernels were given."""
    lib = _lib.load()
    B = (im if im is not None else S).shape[0]
    dev = (im if im is not None else S).device
    D = im.shape[1] if im is not None else 1
    out = {'M': torch.empty((B, B), dtype=torch.float32, device=dev) if flags & (HEAD_MATCH_HINGE | HEAD_LISTNET) else None,
           'terms': torch.empty(3, dtype=torch.float32, device=dev),             # [matching, alignment, listnet]
           'total': torch.empty((), dtype=torch.float32, device=dev)}
    f32 = dict(dtype=torch.float32, device=dev)
    # a head with weight 0 is computed for its logged value only (the distillation term before distill_epoch,
    # alad_model.py:442-444): no gradient matrix is produced for it, and the kernels keep it out of the total
    out['dMh'] = torch.empty((B, B), **f32) if (want_grads and flags & HEAD_MATCH_HINGE and weights[0] != 0) else None
    out['dMl'] = torch.empty((B, B), **f32) if (want_grads and flags & HEAD_LISTNET and weights[2] != 0) else None
    out['dS'] = torch.empty((B, B), **f32) if (want_grads and flags & HEAD_ALIGN_HINGE) else None
    out['pairs'] = out['table_ws'] = out['sets'] = None
    if (align is not None and out['dS'] is not None and max_violation and _pair_kernel_covers(align[4][0])
            and not align[4][0].split):
        im_set, s_seq, im_len_t, s_len_t, packed = align
        geom = packed[0]
        im_c, s_c = _rows_inner_contig(im_set), _rows_inner_contig(s_seq)
        out['table_ws'] = torch.empty(lib.aladin_align_bwd_workspace_bytes(C.byref(geom), 0), dtype=torch.uint8, device=dev)
        out['sets'] = (im_c, s_c)
        ws = _workspace(lib.aladin_heads_small_workspace_bytes(B), dev)
        vi, vs, pk = _set_view(im_c, im_len_t), _set_view(s_c, s_len_t), _packed_struct(*packed[1:5])
        _lib.check(lib.aladin_heads_small_fwd_argmax(_ptr(im), _ld(im) if im is not None else 0, _ptr(s), _ld(s) if s is not None else 0,
                                                     _ptr(S), _ld(S), D, float(margin), int(flags), float(temperature), float(eps),
                                                     float(weights[0]), float(weights[1]), float(weights[2]), _ptr(out['M']),
                                                     _ptr(out['terms']), _ptr(out['total']), _ptr(out['dMh']), _ptr(out['dMl']),
                                                     _ptr(out['dS']), _ptr(ws), C.byref(vi), C.byref(vs), C.byref(geom), C.byref(pk),
                                                     _ptr(out['table_ws']), _stream()), 'heads_small_fwd_argmax')
        return out
    if want_pairs and out['dS'] is not None:
        out['pairs'] = (torch.empty(B * B, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev))
    ws = _workspace(lib.aladin_heads_small_workspace_bytes(B), dev)
    _lib.check(lib.aladin_heads_small_fwd(_ptr(im), _ld(im) if im is not None else 0, _ptr(s), _ld(s) if s is not None else 0,
                                          _ptr(S), _ld(S) if S is not None else 0, B, D, float(margin), int(bool(max_violation)),
                                          int(flags), float(temperature), float(eps), float(weights[0]), float(weights[1]),
                                          float(weights[2]), _ptr(out['M']), _ptr(out['terms']), _ptr(out['total']),
                                          _ptr(out['dMh']), _ptr(out['dMl']), _ptr(out['dS']),
                                          _ptr(out['pairs'][0] if out['pairs'] else None),
                                          _ptr(out['pairs'][1] if out['pairs'] else None), _ptr(ws), _stream()), 'heads_small_fwd')
    return out


class _SmallMatchDistill(torch.autograd.Function):
    """(hinge loss on M, listnet loss of M against the teacher, M) with M = im @ s.T, in two forward launches and
    one backward launch (csrc/small_batch.hip).  Either loss may be switched off (returns a zero scalar)."""

    @staticmethod
    def forward(ctx, im, s, teacher, margin, max_violation, want_hinge, temperature, eps):
        im = im if im.stride(1) == 1 else im.contiguous()
        s = s if s.stride(1) == 1 else s.contiguous()
        t = None
        if teacher is not None:
            t = teacher if teacher.stride(1) == 1 else teacher.contiguous()
        flags = (HEAD_MATCH_HINGE if want_hinge else 0) | (HEAD_LISTNET if t is not None else 0)
        need = any(ctx.needs_input_grad[:2])
        o = _heads_small_fwd(im, s, t, margin, max_violation, flags, temperature, eps, (1.0, 1.0, 1.0), need, False)
        ctx.save_for_backward(im, s, o['dMh'], o['dMl'])
        ctx.set_materialize_grads(False)
        return o['terms'][0], o['terms'][2], o['M']

    @staticmethod
    def backward(ctx, g_h, g_l, g_M):
        im, s, dMh, dMl = ctx.saved_tensors
        if (g_h is None or dMh is None) and (g_l is None or dMl is None) and g_M is None:
            return (None,) * 8
        B, D = im.shape
        d_im = torch.empty((B, D), dtype=torch.float32, device=im.device) if ctx.needs_input_grad[0] else None
        d_s = torch.empty((B, D), dtype=torch.float32, device=im.device) if ctx.needs_input_grad[1] else None
        gh = g_h.to(torch.float32).contiguous() if (g_h is not None and dMh is not None) else None
        gl = g_l.to(torch.float32).contiguous() if (g_l is not None and dMl is not None) else None
        gM = (g_M if g_M.stride(1) == 1 else g_M.contiguous()) if g_M is not None else None
        _lib.check(_lib.load().aladin_heads_small_bwd(_ptr(im), _ld(im), _ptr(s), _ld(s), B, D,
                                                      _ptr(dMh if gh is not None else None), _ptr(gh), 1.0,
                                                      _ptr(dMl if gl is not None else None), _ptr(gl), 1.0, _ptr(gM),
                                                      _ld(gM) if gM is not None else 0, _ptr(None), 0.0, _ptr(None),
                                                      _ptr(d_im), _ptr(d_s), _stream()), 'heads_small_bwd')
        return d_im, d_s, None, None, None, None, None, None


class _SmallHeads(torch.autograd.Function):
    """The whole loss-head step at B <= 64 as ONE autograd node: alignment scores (pack, side GEMM, score kernel),
    then the three heads and their fixed-weight sum (alad_model.py:450-453) in two launches; backward = one launch for
    the matching side + the two alignment backward kernels.  No element-wise glue kernels at all.
    Returns (total, terms[3] = matching / alignment / distillation, S, M); only `total` is differentiable."""

    @staticmethod
    def forward(ctx, img_emb, cap_emb, im, s, im_len_t, s_len_t, margin, max_violation, flags, weights, temperature, eps):
        need_sets = any(ctx.needs_input_grad[2:4])
        need_embs = any(ctx.needs_input_grad[0:2])
        S, packed = None, None
        if flags & (HEAD_ALIGN_HINGE | HEAD_LISTNET):
            if need_sets and flags & HEAD_ALIGN_HINGE:
                _check_backward_supported(im, s, 0, 2)
            S, packed = _align_forward(im, s, im_len_t, s_len_t)
        a = b = None
        if flags & (HEAD_MATCH_HINGE | HEAD_LISTNET):
            a = img_emb if img_emb.stride(1) == 1 else img_emb.contiguous()
            b = cap_emb if cap_emb.stride(1) == 1 else cap_emb.contiguous()
        align = (im, s, im_len_t, s_len_t, packed) if (packed is not None and need_sets and flags & HEAD_ALIGN_HINGE) else None
        o = _heads_small_fwd(a, b, S, margin, max_violation, flags, temperature, eps, weights, need_sets or need_embs, True, align)
        if o['sets'] is not None:
            im, s = o['sets']
        ctx.flags, ctx.weights = flags, weights
        ctx.geom = packed[0] if packed is not None else None
        ctx.pairs = o['pairs']
        pk = packed[1:] if packed is not None else (None, None, None, None)
        ctx.save_for_backward(a, b, im, s, im_len_t, s_len_t, pk[0], pk[1], pk[2], pk[3], o['dMh'], o['dMl'], o['dS'], o['table_ws'])
        ctx.set_materialize_grads(False)
        terms = o['terms']
        ctx.mark_non_differentiable(*[t for t in (terms, S, o['M']) if t is not None])       # one call: it replaces the set
        return o['total'], terms, S, o['M']

    @staticmethod
    def backward(ctx, g_total, _g_terms, _g_S, _g_M):
        if g_total is None:
            return (None,) * 12
        a, b, im, s, im_len_t, s_len_t, xm, xe, y, rnorm, dMh, dMl, dS, table_ws = ctx.saved_tensors
        flags, w = ctx.flags, ctx.weights
        g = g_total.to(torch.float32).contiguous()
        d_a = d_b = d_im = d_s = None
        scale = torch.empty(1, dtype=torch.float32, device=g.device) if dS is not None else None
        if a is not None:
            B, D = a.shape
            d_a = torch.empty((B, D), dtype=torch.float32, device=a.device) if ctx.needs_input_grad[0] else None
            d_b = torch.empty((B, D), dtype=torch.float32, device=a.device) if ctx.needs_input_grad[1] else None
            _lib.check(_lib.load().aladin_heads_small_bwd(_ptr(a), _ld(a), _ptr(b), _ld(b), B, D, _ptr(dMh), _ptr(g), float(w[0]),
                                                          _ptr(dMl), _ptr(g), float(w[2]), _ptr(None), 0, _ptr(g), float(w[1]),
                                                          _ptr(scale), _ptr(d_a), _ptr(d_b), _stream()), 'heads_small_bwd')
        elif scale is not None:
            scale = g * float(w[1])
        if dS is not None and any(ctx.needs_input_grad[2:4]):
            if table_ws is not None:
                d_im, d_s = _triplet_backward(im, s, im_len_t, s_len_t, ctx.geom, _packed_struct(xm, xe, y, rnorm), dS, table_ws, scale,
                                              base_workspace=True)
            else:
                d_im, d_s = _align_backward(im, s, im_len_t, s_len_t, dS, gscale=scale, packed=(ctx.geom, xm, xe, y, rnorm), pairs=ctx.pairs)
        return d_a, d_b, d_im, d_s, None, None, None, None, None, None, None, None


class _BigHeads(torch.autograd.Function):
    """_SmallHeads for B > 64: the same single autograd node over the general kernels -- alignment scores + fused hinge,
    exact-fp32 matching GEMM, hinge and ListNet on it, aladin_loss_total for the weighted sum; backward =
    aladin_grad_combine (upstream gradient x weights x dLoss/dM, and the alignment backward's scale) + two GEMMs + the
    alignment backward.  No element-wise torch kernels."""

    @staticmethod
    def forward(ctx, img_emb, cap_emb, im, s, im_len_t, s_len_t, margin, max_violation, flags, weights, temperature, eps):
        lib = _lib.load()
        need_sets = any(ctx.needs_input_grad[2:4])
        need_embs = any(ctx.needs_input_grad[0:2])
        dev = img_emb.device
        B = img_emb.shape[0]
        terms = torch.empty(3, dtype=torch.float32, device=dev)          # slots of absent heads are never read
        S = packed = dS = pairs = table_ws = buf = None
        dense = False
        ctx.fill, _FILL_HINT[0] = _FILL_HINT[0], None
        ctx.offs = None
        if flags & (HEAD_ALIGN_HINGE | HEAD_LISTNET):
            if need_sets and flags & HEAD_ALIGN_HINGE:
                _check_backward_supported(im, s, 0, 2)
            fused = (_triplet_forward(im, s, im_len_t, s_len_t, margin, loss_out=terms[1:2])
                     if (flags & HEAD_ALIGN_HINGE) and need_sets and max_violation else None)
            if fused is not None:                    # the alignment head's whole forward in one library call
                _, S, (im, s, geom_f, buf, dS, table_ws, ctx.offs) = fused
                packed = (geom_f, None, None, None, None)
            else:
                S, packed = _align_forward(im, s, im_len_t, s_len_t, norms=need_sets)
                if flags & HEAD_ALIGN_HINGE:
                    _, dS, pairs = _hinge_raw(S, margin, max_violation, need_sets, want_pairs=True, loss_out=terms[1:2])
                    if need_sets and not max_violation:                 # sum of violations: the dense backward while dS is dense
                        dense = _density_probe.step(pairs[1], B * B)
        a = b = M = dMh = dMl = None
        if flags & (HEAD_MATCH_HINGE | HEAD_LISTNET):
            a = img_emb if img_emb.stride(1) == 1 else img_emb.contiguous()
            b = cap_emb if cap_emb.stride(1) == 1 else cap_emb.contiguous()
            M = torch.empty((B, B), dtype=torch.float32, device=dev)
            _sgemm(B, B, a.shape[1], a, a.stride(0), a.stride(1), b, b.stride(1), b.stride(0), M)
            if flags & HEAD_MATCH_HINGE:
                _, dMh, _ = _hinge_raw(M, margin, max_violation, need_embs, loss_out=terms[0:1])
            if flags & HEAD_LISTNET:
                dMl = torch.empty((B, B), dtype=torch.float32, device=dev) if (need_embs and weights[2] != 0) else None
                ws = _workspace(lib.aladin_listnet_workspace_bytes(B), dev)
                _lib.check(lib.aladin_listnet_fwd_bwd(_ptr(S), _ld(S), _ptr(M), _ld(M), B, float(temperature), float(eps),
                                                      C.c_void_p(terms.data_ptr() + 8), _ptr(dMl), _ptr(ws), _stream()),
                           'listnet_fwd_bwd')
        total = torch.empty((), dtype=torch.float32, device=dev)
        tp = terms.data_ptr()
        _lib.check(lib.aladin_loss_total(C.c_void_p(tp) if flags & HEAD_MATCH_HINGE else C.c_void_p(0), float(weights[0]),
                                         C.c_void_p(tp + 4) if flags & HEAD_ALIGN_HINGE else C.c_void_p(0), float(weights[1]),
                                         C.c_void_p(tp + 8) if flags & HEAD_LISTNET else C.c_void_p(0), float(weights[2]),
                                         _ptr(total), _stream()), 'loss_total')
        ctx.flags, ctx.weights = flags, weights
        ctx.geom = packed[0] if packed is not None else None
        ctx.pairs = pairs
        ctx.dense = dense
        pk = packed[1:] if packed is not None else (None, None, None, None)
        ctx.save_for_backward(a, b, im, s, im_len_t, s_len_t, pk[0], pk[1], pk[2], pk[3], dMh, dMl, dS, table_ws, buf)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(*[t for t in (terms, S, M) if t is not None])
        return total, terms, S, M

    @staticmethod
    def backward(ctx, g_total, _g_terms, _g_S, _g_M):
        if g_total is None:
            return (None,) * 12
        lib = _lib.load()
        a, b, im, s, im_len_t, s_len_t, xm, xe, y, rnorm, dMh, dMl, dS, table_ws, buf = ctx.saved_tensors
        w = ctx.weights
        g = g_total.to(torch.float32).contiguous()
        dev = g.device
        d_a = d_b = d_im = d_s = None
        want_m = (dMh is not None or dMl is not None) and any(ctx.needs_input_grad[0:2])
        want_a = dS is not None and any(ctx.needs_input_grad[2:4])
        C_tot = torch.empty_like(dMh if dMh is not None else dMl) if want_m else None
        scale = torch.empty(1, dtype=torch.float32, device=dev) if want_a else None
        if want_m or want_a:
            n = C_tot.numel() if C_tot is not None else 0
            _lib.check(lib.aladin_grad_combine(n, _ptr(g), float(w[0]), _ptr(dMh), float(w[2]), _ptr(dMl), _ptr(C_tot), float(w[1]),
                                               _ptr(scale), _stream()), 'grad_combine')
        if want_m:
            B, D = a.shape
            if ctx.needs_input_grad[0]:
                d_a = torch.empty((B, D), dtype=torch.float32, device=dev)
                _sgemm(B, D, B, C_tot, C_tot.stride(0), 1, b, b.stride(0), b.stride(1), d_a)          # C @ cap
            if ctx.needs_input_grad[1]:
                d_b = torch.empty((B, D), dtype=torch.float32, device=dev)
                _sgemm(B, D, B, C_tot, 1, C_tot.stride(0), a, a.stride(0), a.stride(1), d_b)          # C.T @ img
        if want_a and buf is not None:
            d_im, d_s = _triplet_backward(im, s, im_len_t, s_len_t, ctx.geom, _packed_from_buf(buf, ctx.offs), dS, table_ws, scale)
        elif want_a:
            d_im, d_s = _align_backward(im, s, im_len_t, s_len_t, dS, gscale=scale, packed=(ctx.geom, xm, xe, y, rnorm), pairs=ctx.pairs,
                                        dense=ctx.dense, fill=ctx.fill)
        return d_a, d_b, d_im, d_s, None, None, None, None, None, None, None, None


def small_batch_loss_heads(img_emb, cap_emb, im_set, s_seq, im_len, s_len, margin, max_violation, heads, weights,
                           temperature=6.0, eps=1e-10):
    """The loss heads of one training step in a single autograd node (three head launches at B <= SMALL_BATCH_MAX,
    the general kernels above it -- either way no element-wise glue).
    heads: subset of {'matching', 'alignment', 'distillation'}; weights: dict head -> fixed loss weight.
    -> (total = sum_k w_k L_k  [differentiable], terms (3,) = matching / alignment / distillation values, S, M)."""
    flags = (HEAD_MATCH_HINGE if 'matching' in heads else 0) | (HEAD_ALIGN_HINGE if 'alignment' in heads else 0) | \
        (HEAD_LISTNET if 'distillation' in heads else 0)
    if not flags:
        raise ValueError('aladin_amd: no loss head selected')
    _require_gpu(img_emb, cap_emb)
    im_len_t = s_len_t = None
    if flags & (HEAD_ALIGN_HINGE | HEAD_LISTNET):
        im_len_t, s_len_t = _check_sets(im_set, s_seq, im_len, s_len)
        if not (im_set.shape[0] == s_seq.shape[0] == img_emb.shape[0]):
            raise ValueError('aladin_amd: the loss heads need one image set, one caption and one embedding pair per sample')
        if torch.is_grad_enabled() and (im_set.requires_grad or s_seq.requires_grad):
            from .ops import _pad_features
            im_set, s_seq = _pad_features(im_set, s_seq)          # D % 4 != 0: zero features, dropped again by autograd
    w = (float(weights.get('matching', 0.0)), float(weights.get('alignment', 0.0)), float(weights.get('distillation', 0.0)))
    node = _SmallHeads if img_emb.shape[0] <= SMALL_BATCH_MAX else _BigHeads
    _FILL_HINT[0] = _caption_fill(s_len, s_seq.shape[1]) if (node is _BigHeads and not max_violation and flags & HEAD_ALIGN_HINGE) else None
    return node.apply(img_emb, cap_emb, im_set, s_seq, im_len_t, s_len_t, margin, max_violation, flags, w, temperature, eps)


loss_heads = small_batch_loss_heads            # the single-node step at any batch size


class _MatchHinge(torch.autograd.Function):
    """(hinge loss on M, M) with M = im @ s.T (alad/loss.py:8-11 + :42-67) as ONE autograd node at any batch size: forward = the
    exact-fp32 GEMM + the fused hinge, backward = aladin_grad_combine (g_loss * dLoss/dM + g_M in one launch) + two GEMMs.  Both
    outputs are differentiable, as ContrastiveLoss(return_similarity_mat=True)'s are in the reference."""

    @staticmethod
    def forward(ctx, im, s, margin, max_violation):
        a = im if im.stride(1) == 1 else im.contiguous()
        b = s if s.stride(1) == 1 else s.contiguous()
        B = a.shape[0]
        M = torch.empty((B, B), dtype=torch.float32, device=a.device)
        _sgemm(B, B, a.shape[1], a, a.stride(0), a.stride(1), b, b.stride(1), b.stride(0), M)
        need = any(ctx.needs_input_grad[:2])
        loss, dM, _ = _hinge_raw(M, margin, max_violation, need)
        ctx.save_for_backward(a, b, dM)
        ctx.set_materialize_grads(False)
        return loss, M

    @staticmethod
    def backward(ctx, g_loss, g_M):
        a, b, dM = ctx.saved_tensors
        if (g_loss is None or dM is None) and g_M is None:
            return None, None, None, None
        lib = _lib.load()
        B, D = a.shape
        if g_loss is not None and dM is not None:
            g = g_loss.to(torch.float32).contiguous()
            gM = g_M.contiguous() if g_M is not None else None
            C_tot = torch.empty_like(dM)
            # C = g * (1 * dM) (+ g_M: the second matrix slot with weight 1 / g -- not expressible, so add it separately when present)
            _lib.check(lib.aladin_grad_combine(C_tot.numel(), _ptr(g), 1.0, _ptr(dM), 0.0, _ptr(None), _ptr(C_tot), 0.0, _ptr(None),
                                               _stream()), 'grad_combine')
            if gM is not None:
                C_tot = C_tot + gM
        else:
            C_tot = g_M.contiguous()
        d_a = d_b = None
        if ctx.needs_input_grad[0]:
            d_a = torch.empty((B, D), dtype=torch.float32, device=a.device)
            _sgemm(B, D, B, C_tot, C_tot.stride(0), 1, b, b.stride(0), b.stride(1), d_a)          # C @ cap
        if ctx.needs_input_grad[1]:
            d_b = torch.empty((B, D), dtype=torch.float32, device=a.device)
            _sgemm(B, D, B, C_tot, 1, C_tot.stride(0), a, a.stride(0), a.stride(1), d_b)          # C.T @ img
        return d_a, d_b, None, None


def match_hinge(im, s, margin, max_violation):
    """(loss, M) of ContrastiveLoss(measure='dot') in one autograd node -- the matching head as alad_model.py:380 calls it
    every step.  B <= SMALL_BATCH_MAX: the small-batch kernels (two launches forward, one backward); above: GEMM + fused hinge."""
    _require_gpu(im, s)
    if im.dim() != 2 or s.dim() != 2 or im.shape != s.shape:
        raise ValueError('aladin_amd: the contrastive loss needs a square score matrix: two (B, D) embedding matrices of equal shape '
                         '(the reference fails in diag/expand_as, alad/loss.py:43-45); got %s and %s' % (tuple(im.shape), tuple(s.shape)))
    if im.shape[0] <= SMALL_BATCH_MAX:
        loss, _, M = _SmallMatchDistill.apply(im, s, None, margin, max_violation, True, 6.0, 1e-10)
        return loss, M
    return _MatchHinge.apply(im, s, margin, max_violation)


def small_batch_match_distill(im, s, teacher, margin, max_violation, want_hinge=True, temperature=6.0, eps=1e-10):
    """-> (hinge_loss, listnet_loss, M) for B <= SMALL_BATCH_MAX unit-norm global embeddings im, s (B, D):
    M = im @ s.T (alad/loss.py:8-11), the VSE++ hinge on it (:42-67, if want_hinge) and the ListNet distillation from
    `teacher` (:427-445, detached; None = no distillation) -- two launches per step instead of about twelve."""
    _require_gpu(im, s)
    if im.dim() != 2 or s.dim() != 2 or im.shape != s.shape:
        raise ValueError('aladin_amd: two (B, D) embedding matrices of equal shape expected')
    if im.shape[0] > SMALL_BATCH_MAX:
        raise ValueError('aladin_amd: small_batch_match_distill takes B <= %d' % SMALL_BATCH_MAX)
    if teacher is not None:
        _require_gpu(teacher)
        if tuple(teacher.shape) != (im.shape[0], im.shape[0]):
            raise ValueError('aladin_amd: teacher scores must be (B, B)')
        teacher = teacher.detach()
    if not want_hinge and teacher is None:                    # scores only: the plain differentiable dot-product node
        return im.new_zeros(()), im.new_zeros(()), dot_scores(im, s)
    return _SmallMatchDistill.apply(im, s, teacher, margin, max_violation, want_hinge, temperature, eps)

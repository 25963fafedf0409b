"""Score-matrix losses and the exact-fp32 products over the C ABI: the VSE++ hinge, ListNet and the other distillation modes,
order / dot similarities, l2norm (reference alad/loss.py:8-67, 359-447, alad/utils.py:134-139).  Autograd plumbing only."""
import ctypes as C

import torch

from . import _lib
from ._ops_common import _RAW_STREAM, _stream, _ptr, _require_gpu, _rows_inner_contig, _LEN_CACHE, lengths_tensor, _ld, _workspace


def _hinge_raw(scores, margin, max_violation, want_grad, want_pairs=False, loss_out=None):
    """-> (loss, dS or None, pairs or None); pairs = (int32 list of non-zero i*B+j, int32 count).
    loss_out: a one-element float32 view the kernel writes the loss into (instead of a fresh scalar)."""
    lib = _lib.load()
    B = scores.shape[0]
    sc = scores if scores.stride(1) == 1 else scores.contiguous()
    dev = scores.device
    loss = loss_out if loss_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    dS = torch.empty((B, B), dtype=torch.float32, device=dev) if want_grad else None
    ws = _workspace(lib.aladin_hinge_workspace_bytes(B), dev)
    pairs = None
    if want_grad and want_pairs:
        pairs = (torch.empty(B * B, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev))
    _lib.check(lib.aladin_hinge_fused(_ptr(sc), _ld(sc), B, float(margin), int(bool(max_violation)), _ptr(loss),
                                      _ptr(dS), _ptr(pairs[0] if pairs else None), _ptr(pairs[1] if pairs else None),
                                      _ptr(ws), _stream()), 'hinge_fused')
    return loss, dS, pairs


# ------------------------------------------------------------------------------------------------
# hinge / listnet
# ------------------------------------------------------------------------------------------------
class _Hinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, margin, max_violation):
        loss, ctx.dS, _ = _hinge_raw(scores, margin, max_violation, ctx.needs_input_grad[0])
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.dS * g if ctx.dS is not None else None), None, None


def hinge_loss(scores, margin, max_violation):
    """VSE++ hinge on a square score matrix; replaces reference alad/loss.py:42-67."""
    _require_gpu(scores)
    if scores.dim() != 2 or scores.shape[0] != scores.shape[1]:
        raise ValueError('aladin_amd: the contrastive loss needs a square score matrix, got %s '
                         '(the reference fails in diag/expand_as, alad/loss.py:43-45)' % (tuple(scores.shape),))
    return _Hinge.apply(scores, margin, max_violation)


class _ListNet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, teacher, student, temperature, eps):
        lib = _lib.load()
        B = student.shape[0]
        t = teacher if teacher.stride(1) == 1 else teacher.contiguous()
        m = student if student.stride(1) == 1 else student.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=student.device)
        dM = torch.empty((B, B), dtype=torch.float32, device=student.device) if ctx.needs_input_grad[1] else None
        ws = _workspace(lib.aladin_listnet_workspace_bytes(B), student.device)
        _lib.check(lib.aladin_listnet_fwd_bwd(_ptr(t), _ld(t), _ptr(m), _ld(m), B, float(temperature),
                                              float(eps), _ptr(loss), _ptr(dM), _ptr(ws), _stream()), 'listnet_fwd_bwd')
        ctx.dM = dM
        return loss

    @staticmethod
    def backward(ctx, g):
        return None, (ctx.dM * g if ctx.dM is not None else None), None, None


def listnet_loss(teacher_scores, student_scores, temperature=6.0, eps=1e-10):
    """ListNet distillation; replaces reference alad/loss.py:427-445 (teacher detached, :370)."""
    _require_gpu(teacher_scores, student_scores)
    if teacher_scores.shape != student_scores.shape or student_scores.dim() != 2 \
            or student_scores.shape[0] != student_scores.shape[1]:
        raise ValueError('aladin_amd: listnet needs two square score matrices of equal shape')
    return _ListNet.apply(teacher_scores.detach(), student_scores, temperature, eps)


class _DistillMode(torch.autograd.Function):
    """mse / contrastive / ordinal distillation: forward computes loss and d student in one call."""

    @staticmethod
    def forward(ctx, teacher, student, wb, mode, margin, threshold, stride):
        lib = _lib.load()
        B = student.shape[0]
        t = teacher if teacher.stride(1) == 1 else teacher.contiguous()
        m = student if student.stride(1) == 1 else student.contiguous()
        dev = student.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dM = torch.empty((B, B), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        ws = _workspace(lib.aladin_distill_workspace_bytes(B), dev)
        ctx.dwb = None
        if mode == 'mse':
            w = wb.detach().to(torch.float32).contiguous()
            ctx.dwb = torch.empty(2, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
            _lib.check(lib.aladin_distill_mse_fwd_bwd(_ptr(t), _ld(t), _ptr(m), _ld(m), B, _ptr(w), _ptr(loss), _ptr(dM),
                                                      _ptr(ctx.dwb), _ptr(ws), _stream()), 'distill_mse_fwd_bwd')
        elif mode == 'contrastive':
            _lib.check(lib.aladin_distill_contrastive_fwd_bwd(_ptr(t), _ld(t), _ptr(m), _ld(m), B, float(margin), _ptr(loss),
                                                              _ptr(dM), _ptr(ws), _stream()), 'distill_contrastive_fwd_bwd')
        else:
            _lib.check(lib.aladin_distill_ordinal_fwd_bwd(_ptr(t), _ld(t), _ptr(m), _ld(m), B, float(margin),
                                                          float(threshold), int(stride), _ptr(loss), _ptr(dM), _ptr(ws),
                                                          _stream()), 'distill_ordinal_fwd_bwd')
        ctx.dM = dM
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None, ctx.dM * g if ctx.dM is not None else None, ctx.dwb * g if ctx.dwb is not None else None,
                None, None, None, None)


def distillation_loss(teacher_scores, student_scores, mode, margin=0.2, threshold=0.1, stride=3, wb=None):
    """DistillationLoss modes 'mse' / 'contrastive' / 'ordinal'; replaces reference alad/loss.py:371-425
    (teacher detached, :370).  ``wb`` is the learnable (2,) pair of the 'mse' mode (:366)."""
    _require_gpu(teacher_scores, student_scores)
    if mode not in ('mse', 'contrastive', 'ordinal'):
        raise ValueError('aladin_amd: unknown distillation mode %r' % (mode,))
    if teacher_scores.shape != student_scores.shape or student_scores.dim() != 2 \
            or student_scores.shape[0] != student_scores.shape[1]:
        raise ValueError('aladin_amd: distillation needs two square score matrices of equal shape')
    if mode == 'mse':
        if wb is None or wb.numel() != 2:
            raise ValueError("aladin_amd: mode 'mse' needs the (2,) parameter wb")
        _require_gpu(wb)
    elif mode == 'ordinal' and not 1 <= int(stride) < student_scores.shape[0]:
        raise ValueError('aladin_amd: ordinal distillation needs 1 <= stride < B')
    return _DistillMode.apply(teacher_scores.detach(), student_scores, wb, mode, margin, threshold, stride)


class _OrderScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s):
        lib = _lib.load()
        a = im if im.stride(1) == 1 else im.contiguous()
        b = s if s.stride(1) == 1 else s.contiguous()
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
        _lib.check(lib.aladin_order_sim_fwd(_ptr(a), a.stride(0), _ptr(b), b.stride(0), a.shape[0], b.shape[0], a.shape[1],
                                            _ptr(out), out.stride(0), _stream()), 'order_sim_fwd')
        ctx.save_for_backward(a, b, out)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b, out = ctx.saved_tensors
        g = g.contiguous()
        d_im = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        d_s = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.check(lib.aladin_order_sim_bwd(_ptr(a), a.stride(0), _ptr(b), b.stride(0), a.shape[0], b.shape[0], a.shape[1],
                                            _ptr(g), _ld(g), _ptr(out), out.stride(0), _ptr(d_im),
                                            d_im.stride(0) if d_im is not None else 0, _ptr(d_s),
                                            d_s.stride(0) if d_s is not None else 0, _stream()), 'order_sim_bwd')
        return d_im, d_s


def order_scores(im, s):
    """-||max(s_j - im_i, 0)||; replaces order_sim, reference alad/loss.py:20-26."""
    _require_gpu(im, s)
    if im.dim() != 2 or s.dim() != 2 or im.shape[1] != s.shape[1]:
        raise ValueError('aladin_amd: (Bi,D) and (Bc,D) embeddings expected')
    return _OrderScores.apply(im, s)


# ------------------------------------------------------------------------------------------------
# dot-product scores (matching head)
# ------------------------------------------------------------------------------------------------
def _sgemm(M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, out):
    _lib.check(_lib.load().aladin_sgemm_strided(M, N, K, _ptr(A), a_rs, a_cs, _ptr(B), b_rs, b_cs, _ptr(out),
                                                out.stride(0), _stream()), 'sgemm_strided')
    return out


class _DotScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s):
        ctx.save_for_backward(im, s)
        out = torch.empty((im.shape[0], s.shape[0]), dtype=torch.float32, device=im.device)
        # C[m][n] = sum_k im[m,k] * s[n,k]
        return _sgemm(im.shape[0], s.shape[0], im.shape[1], im, im.stride(0), im.stride(1), s, s.stride(1), s.stride(0), out)

    @staticmethod
    def backward(ctx, dM):
        im, s = ctx.saved_tensors
        dM = dM.contiguous()
        Bi, Bc, D = im.shape[0], s.shape[0], im.shape[1]
        d_im = torch.empty((Bi, D), dtype=torch.float32, device=im.device)
        d_s = torch.empty((Bc, D), dtype=torch.float32, device=im.device)
        _sgemm(Bi, D, Bc, dM, dM.stride(0), 1, s, s.stride(0), s.stride(1), d_im)        # dM @ s
        _sgemm(Bc, D, Bi, dM, 1, dM.stride(0), im, im.stride(0), im.stride(1), d_s)      # dM.T @ im
        return d_im, d_s


def dot_scores(im, s):
    """im @ s.T in exact fp32 on the MFMA; replaces dot_sim, reference alad/loss.py:8-11."""
    _require_gpu(im, s)
    if im.dim() != 2 or s.dim() != 2 or im.shape[1] != s.shape[1]:
        raise ValueError('aladin_amd: (Bi,D) and (Bc,D) embeddings expected')
    return _DotScores.apply(im, s)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x if x.stride(1) == 1 else x.contiguous()
        out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().aladin_l2norm_fwd(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(out), _stream()), 'l2norm_fwd')
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g if g.stride(1) == 1 else g.contiguous()
        dx = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().aladin_l2norm_bwd(_ptr(x), x.stride(0), _ptr(g), _ld(g), x.shape[0], x.shape[1], _ptr(dx), _stream()),
                   'l2norm_bwd')
        return dx


def l2norm_rows(x):
    """X / sqrt(sum_dim1 X^2) without eps; replaces l2norm, reference alad/utils.py:134-139 (zero rows -> NaN)."""
    _require_gpu(x)
    if x.dim() != 2 or x.shape[0] < 1 or x.shape[1] < 1:
        raise ValueError('aladin_amd: l2norm expects a non-empty (rows, D) matrix')
    return _L2Norm.apply(x)

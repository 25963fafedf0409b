"""Tensor-level entry points over the C ABI (include/aladin_hip.h): device memory, streams and
autograd plumbing only -- all arithmetic of the hot path happens in the HIP kernels.

Every function requires float32 tensors on an AMD GPU ("cuda" device of PyTorch-ROCm) and raises
otherwise; there is no CPU or eager-PyTorch fallback.
"""
import ctypes as C

import torch

from . import _lib

from ._ops_common import _RAW_STREAM, _stream, _ptr, _require_gpu, _rows_inner_contig, _LEN_CACHE, lengths_tensor, _ld, _workspace
from .ops_losses import _DotScores, _hinge_raw        # (the other names of ops_losses / ops_heads / ops_retrieval: __getattr__ below)


def _pair_kernel_covers(geom):
    """Shapes the fp16 pair kernel of the backward covers (csrc/align_bwd.hip: bwd_pair_argmax16_kernel): a 64-row block per
    pair = the image's 32 / 48 main rows + a window on its side rows, or its 64 main rows; at most 64 padded words."""
    return (geom.mrows in (32, 48) or (geom.mrows == 64 and geom.rem == 0)) and geom.tp16 <= 4


# ------------------------------------------------------------------------------------------------
# alignment scores
# ------------------------------------------------------------------------------------------------
_GEOM_CACHE = {}

# Operand precision of score matrices computed WITHOUT autograd (evaluation, validation, the reference's
# alignment_sim_fn closures -- train.py:495-498, test.py:260-263):
#   'split'  hi/lo fp16 split, three MFMA products per term: scores at the rounding level of the reference's
#            own fp32 bmm, so Recall@K ranks are the reference's (default);
#   'fp16'   the training operands (one fp16 rounding, ~1e-4 on a score): 3x faster, near-ties may swap.
# Differentiable scores always use fp16 operands (north_star's 1e-3 tolerance; the backward re-decides its
# arg-maxima in exact fp32 anyway).
_EVAL_PRECISION = ['split']
E_SCRATCH_LIMIT = 2 << 30       # bytes of side-GEMM scratch per score launch before the sum side is chunked


def set_eval_precision(precision):
    """'split' (rank-exact, default) or 'fp16' for no-grad alignment scores; returns the previous setting."""
    if precision not in ('split', 'fp16'):
        raise ValueError("aladin_amd: eval precision must be 'split' or 'fp16'")
    old = _EVAL_PRECISION[0]
    _EVAL_PRECISION[0] = precision
    return old


# Where the backward's row kernel takes its unit vectors from (tools/experiments/bwd_precision_probe.py on every reference fixture,
# profiles/r05_bwd_precision_probe.txt; the gate is 5e-4 of the largest gradient entry = half of north_star's 1e-3):
#   'fp16'      ALADIN_BWD_PARTNERS_FP16 (default since round 5, VERDICT r4 item 2 iii): the PARTNER rows -- the unit vectors a
#               gradient row is a weighted sum of -- from the forward's packed fp16 operands: worst 4.3e-4 over the fixtures (one
#               fp16 rounding: <= 2^-11 relative per component), row kernel 43.5 -> 34.0 us at B = 256, 302 -> 217 MB;
#   'fp16-own'  + ALADIN_BWD_OWN_ROW_FP16 (item 2 i): the output row's own unit vector and inverse norm from the packed operands too
#               -- the raw fp32 sets are not read by the row kernel at all: 30.4 us.  Same error wherever D >= 128; on the D = 64
#               structured fixture (cosines near 1: the projection term is as large as the gradient) 5.8e-4 -- past the gate, so an opt-in;
#   'exact'     the raw fp32 sets, normalised again in fp32: 3e-7 of the reference's autograd.
# The fp16 modes apply wherever the forward's packed operands (with their inverse norms) reach the backward.
_BWD_PARTNERS = ['fp16']
_BWD_OWN_ROW_FP16 = [False]


def set_backward_precision(mode):
    """'fp16' (default), 'fp16-own' or 'exact' unit vectors in the alignment backward's row step; returns the previous setting."""
    if mode not in ('exact', 'fp16', 'fp16-own'):
        raise ValueError("aladin_amd: backward precision must be 'exact', 'fp16' or 'fp16-own'")
    old = 'exact' if _BWD_PARTNERS[0] == 'exact' else ('fp16-own' if _BWD_OWN_ROW_FP16[0] else 'fp16')
    _BWD_PARTNERS[0] = 'exact' if mode == 'exact' else 'fp16'
    _BWD_OWN_ROW_FP16[0] = mode == 'fp16-own'
    return old


def _bwd_flags(packed):
    """flags word of the backward entry points for this problem: ALADIN_BWD_PARTNERS_FP16 when the setting asks for it and the
    forward's packed operands WITH their inverse norms are at hand (a 5-tuple from _align_forward / pack_sets)."""
    if _BWD_PARTNERS[0] == 'fp16' and packed is not None and len(packed) > 4 and packed[1] is not None and packed[3] is not None \
            and packed[4] is not None and not packed[0].split:
        return _lib.BWD_PARTNERS_FP16 | (_lib.BWD_OWN_ROW_FP16 if _BWD_OWN_ROW_FP16[0] else 0)
    return 0


def _precision_code(precision):
    if precision in (None, 'fp16', _lib.PRECISION_FP16):
        return _lib.PRECISION_FP16
    if precision in ('split', _lib.PRECISION_SPLIT):
        return _lib.PRECISION_SPLIT
    raise ValueError("aladin_amd: precision must be 'fp16' or 'split', got %r" % (precision,))


def align_geometry(Bi, Bc, R, T, D, x_tail=0, y_tail=2, precision=None):
    """Packed layout for a (max-side set Bi x R) x (sum-side set Bc x T) problem; the tails are the
    trailing positions each set drops (images 0, captions 2 -- reference alad/loss.py:87-90)."""
    prec = _precision_code(precision)
    key = (Bi, Bc, R, T, D, x_tail, y_tail, prec)
    g = _GEOM_CACHE.get(key)
    if g is None:
        g = _lib.AlignGeom()
        _lib.check(_lib.load().aladin_align_geometry(Bi, Bc, R, T, D, x_tail, y_tail, prec, C.byref(g)), 'align_geometry')
        if len(_GEOM_CACHE) < 1024:
            _GEOM_CACHE[key] = g
    return g


def _set_view(t, len_t):
    """struct aladin_set of a (B, N, D) tensor with a unit inner stride."""
    return _lib.SetView(t.data_ptr(), t.stride(0), t.stride(1), len_t.data_ptr())


def _grad_view(t):
    return _lib.GradView(t.data_ptr(), t.stride(0), t.stride(1))


def _packed_struct(xm, xe, y, rnorm=None):
    return _lib.Packed(xm.data_ptr() if xm is not None else None, xe.data_ptr() if xe is not None else None,
                       y.data_ptr() if y is not None else None, rnorm.data_ptr() if rnorm is not None else None)


def _rnorm_views(rnorm, geom):
    """(image part, caption part) of the inverse-norm buffer [xm rows | xe rows | y rows]."""
    n_img = int(geom.xm_rows + geom.xe_rows)
    return rnorm[:n_img], rnorm[n_img:]


def pack_images(im, im_len_t, geom, rnorm=None, out=None):
    """(xm, xe): L2-normalised, sliced, length-masked fp16 MFMA operands of the image sets.  rnorm (optional float32 tensor of
    geom.rnorm_bytes / 4 elements): receives the image rows' inverse norms in its [xm | xe] part.  out = (xm, xe): write into
    the caller's buffers (the sharded step packs straight into its all-gather segment)."""
    lib = _lib.load()
    im = _rows_inner_contig(im)
    if out is not None:
        xm, xe = out
    else:
        xm = torch.empty(geom.xm_bytes // 2, dtype=torch.float16, device=im.device)
        xe = torch.empty(max(geom.xe_bytes // 2, 8), dtype=torch.float16, device=im.device)
    v, pk = _set_view(im, im_len_t), _packed_struct(xm, xe, None, rnorm)
    _lib.check(lib.aladin_align_pack(C.byref(v), None, C.byref(geom), C.byref(pk), _stream()), 'align_pack(images)')
    return xm, xe


def pack_captions(s, s_len_t, geom, rnorm=None):
    lib = _lib.load()
    s = _rows_inner_contig(s)
    y = torch.empty(geom.y_bytes // 2, dtype=torch.float16, device=s.device)
    v, pk = _set_view(s, s_len_t), _packed_struct(None, None, y, rnorm)
    _lib.check(lib.aladin_align_pack(None, C.byref(v), C.byref(geom), C.byref(pk), _stream()), 'align_pack(captions)')
    return y


def scores_from_packed(xm, xe, y, geom, out=None, e_scratch=None, reuse_side=False):
    lib = _lib.load()
    S = out if out is not None else torch.empty((geom.Bi, geom.Bc), dtype=torch.float32, device=xm.device)
    e = e_scratch if e_scratch is not None else _workspace(geom.e_bytes, xm.device)
    pk = _packed_struct(xm, xe, y)
    _lib.check(lib.aladin_align_scores(C.byref(pk), C.byref(geom), _ptr(e), _ptr(S), S.stride(0), 1 if reuse_side else 0, _stream()),
               'align_scores')
    return S


def _pad_features(im_set, s_seq):
    """The backward kernels move rows as 16-byte float4 columns (D % 4 == 0).  The reference has no such limit: a set whose
    feature size is not a multiple of 4 is padded with zero features here, OUTSIDE the autograd nodes -- zeros change neither
    a norm nor a dot product, and autograd's own backward of the padding drops the extra gradient columns.  One copy of each set,
    paid by odd feature sizes only (every shipped configuration has D = 768)."""
    D = im_set.shape[-1]
    if D % 4 == 0 or im_set.dim() != 3 or s_seq.dim() != 3 or s_seq.shape[-1] != D:
        return im_set, s_seq
    pad = 4 - D % 4
    return torch.nn.functional.pad(im_set, (0, pad)), torch.nn.functional.pad(s_seq, (0, pad))


def _check_backward_supported(im, s, x_tail, y_tail):
    """The limits of aladin_align_bwd (align_bwd.hip), checked when the FORWARD of a differentiable score
    matrix is requested, so that an unsupported shape fails here and not inside loss.backward()."""
    Bi, R, D = im.shape
    Bc, T, _ = s.shape
    if D % 4 != 0 or D > 1024:
        raise ValueError('aladin_amd: differentiable alignment scores need D <= 1024 (got D=%d; the public entry points pad a '
                         'feature size that is not a multiple of 4); score under torch.no_grad()' % D)
    if R - 1 - x_tail > 96 or T - 1 - y_tail > 96:          # = the packed geometry's own limits (aladin_align_geometry)
        raise ValueError('aladin_amd: alignment scores support at most 96 scored positions per set '
                         '(got %d on the max side, %d on the sum side)' % (R - 1 - x_tail, T - 1 - y_tail))


def pack_sets(im, s, im_len_t, s_len_t, geom, norms=True):
    """Both sets in one launch -> (geom, xm, xe, y, rnorm): the `packed` tuple the score and backward functions take
    (rnorm None when norms=False or the operands are split)."""
    dev = im.device
    xm = torch.empty(geom.xm_bytes // 2, dtype=torch.float16, device=dev)
    xe = torch.empty(max(geom.xe_bytes // 2, 8), dtype=torch.float16, device=dev)
    y = torch.empty(geom.y_bytes // 2, dtype=torch.float16, device=dev)
    rnorm = torch.empty(geom.rnorm_bytes // 4, dtype=torch.float32, device=dev) if norms and not geom.split else None
    vi, vs, pk = _set_view(im, im_len_t), _set_view(s, s_len_t), _packed_struct(xm, xe, y, rnorm)
    _lib.check(_lib.load().aladin_align_pack(C.byref(vi), C.byref(vs), C.byref(geom), C.byref(pk), _stream()), 'align_pack')
    return geom, xm, xe, y, rnorm


def _align_forward(im, s, im_len_t, s_len_t, x_tail=0, y_tail=2, precision=None, norms=True):
    """-> (S, packed) where packed = (geom, xm, xe, y, rnorm) is kept for the backward pass.
    `im` is the max-side set, `s` the sum-side set (images / captions for 'MrSw')."""
    Bi, R, D = im.shape
    Bc, T, D2 = s.shape
    if D != D2:
        raise ValueError('aladin_amd: feature sizes differ (%d vs %d)' % (D, D2))
    geom = align_geometry(Bi, Bc, R, T, D, x_tail, y_tail, precision)
    packed = pack_sets(_rows_inner_contig(im), _rows_inner_contig(s), im_len_t, s_len_t, geom, norms=norms)
    return scores_from_packed(packed[1], packed[2], packed[3], geom), packed


def _grad_like(x):
    """Gradient buffer in x's own layout when x is a dense permutation with a unit inner stride (the model's
    (S,B,D)->(B,S,D) views): autograd then hands it to the leaf without a re-layout copy.  Contiguous otherwise."""
    if x.stride(-1) == 1 and torch.ops.aten.is_non_overlapping_and_dense(x) and all(st % 4 == 0 for st in x.stride()[:-1]):
        return torch.empty_like(x, dtype=torch.float32)
    return torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)


DENSE_MIN_FRACTION = 0.4     # of the pairs carrying a gradient: below, the list path (cost ~ 3.2 ms x fraction at B = 256) beats the dense one (~1.4 ms)
DENSE_MIN_PAIRS = 1 << 14    # below: the per-pair kernel is launch-bound and already faster (B = 64: 0.20 vs 0.26 ms)
DENSE_ROWS_GEMM = True       # False: ALADIN_BWD_DENSE_GATHER -- the dense table with the per-row gather (bit-identical to the list path)
DENSE_BACKWARD = True        # False: always one workgroup per gradient-carrying pair (the A/B switch of tests/ and tools/experiments/bench_dense_ds.py)


class _DensityProbe:
    """How dense is dloss/dS of the sum-of-violations hinge?  Early in training nearly every pair violates the margin, late
    few do -- and only the device knows (the hinge kernel's pair count).  Reading it at once would stall the host, so each
    step's count leaves by ONE asynchronous copy and the choice between the dense and the list backward at step n follows
    the count of step n - LAG (the density drifts slowly).  The lag is FIXED, not "whatever has arrived": the two paths sum
    in different orders, and which one runs must not depend on timing (bitwise reproducible runs); waiting for a copy
    issued LAG steps ago costs nothing unless the host is more than LAG steps ahead of the device.
    Before step LAG: `unknown` (dense for the fused sum-of-violations node, the list path for a generic gradient)."""
    LAG = 2

    def __init__(self, unknown=True):
        self._hist = []                                     # [buffer, event, total] of the last LAG + 1 recorded steps, oldest first
        self.fraction = None
        self.unknown = unknown

    def step(self, count, total):
        """Record this step's count; -> dense? for this step."""
        if torch.cuda.is_current_stream_capturing():        # no host copies inside a graph: keep the last answer
            return self.dense()
        if len(self._hist) >= self.LAG:
            buf, ev, tot = self._hist[-self.LAG]
            ev.synchronize()
            self.fraction = int(buf[0]) / tot
        if len(self._hist) > self.LAG:
            slot = self._hist.pop(0)
        else:
            slot = [torch.empty(1, dtype=count.dtype).pin_memory(), torch.cuda.Event(), 1.0]
        if slot[0].dtype != count.dtype:
            slot[0] = torch.empty(1, dtype=count.dtype).pin_memory()
        slot[0].copy_(count.reshape(1), non_blocking=True)
        slot[1].record()
        slot[2] = float(total)
        self._hist.append(slot)
        return self.dense()

    def dense(self):
        return self.unknown if self.fraction is None else self.fraction >= DENSE_MIN_FRACTION

    def newest(self):
        """The fraction of the most recent recorded step (waits for it; for tests / tools)."""
        if not self._hist:
            return None
        buf, ev, tot = self._hist[-1]
        ev.synchronize()
        return int(buf[0]) / tot


_density_probe = _DensityProbe()
# a gradient arriving on a score matrix from OUTSIDE the fused nodes (ops.alignment_scores + any loss: the 'MwSr' / 'symm'
# poolings, a caller's own criterion): its non-zeros are counted (one small launch) and, as above, steer the backward of
# the same shape LAG steps later; until then the list path (the usual dS is the hardest-negative hinge's: <= 3B entries)
_generic_probes = {}
_LAST_BWD_FLAGS = [0]
DENSE_GEMM_FORCE = False     # tests / tools: the GEMM row step whatever the captions' fill


def _caption_fill(s_len, T, y_tail=2):
    """Mean share of a caption's 16-word tiles that holds real words (None: lengths only on the device).  The gather row
    step costs ~ real words, the GEMM one the padded tiles: measured crossover (B = 256, tools/experiments/dense_backward_probe.py fill)
    at a fill of 0.42 for R' = 33 and 0.50 for R' = 50 -- COCO captions (~12 of 35 tokens) are on the gather side."""
    if isinstance(s_len, torch.Tensor):
        return None
    Tq = T - 1 - y_tail
    if Tq < 1 or len(s_len) == 0:
        return None
    tile = (Tq + 15) // 16 * 16
    return sum(min(max(int(v) - 1 - y_tail, 0), Tq) for v in s_len) / float(len(s_len) * tile)


def _gemm_rows_pay(fill, Rq):
    if DENSE_GEMM_FORCE or fill is None:
        return True
    return fill >= min(0.7, max(0.35, 0.42 + 0.08 * (Rq - 33) / 17.0))


_FILL_HINT = [None]          # set by the wrappers just before .apply(), read by the node's forward (same thread, same call)


def _align_backward(im, s, im_len_t, s_len_t, dS, gscale=None, packed=None, pairs=None, x_tails=(0, 2), dense=False, fill=None):
    """aladin_align_bwd.  packed: (geom, xm, xe, y[, rnorm]) of the forward (None: the exact fp32 recompute).
    dense: the caller knows that (almost) every pair carries a gradient (sum-of-violations hinge, a gradient on S):
    ALADIN_BWD_DENSE -- the arg-max table of all pairs from the split-precision tile kernel instead of one workgroup per pair."""
    lib = _lib.load()
    im = _rows_inner_contig(im)
    s = _rows_inner_contig(s)
    dS = dS.contiguous()
    ld_dS = dS.shape[1]            # (not stride(0): a contiguous (1, n) view may report any leading stride)
    Bi, R, D = im.shape
    Bc, T, _ = s.shape
    d_im, d_s = _grad_like(im), _grad_like(s)
    have = packed is not None and packed[1] is not None
    dense_flag = _lib.BWD_DENSE if (dense and DENSE_BACKWARD and have and Bi * Bc >= DENSE_MIN_PAIRS) else 0
    if dense_flag and not (DENSE_ROWS_GEMM and _gemm_rows_pay(fill, R - 1 - packed[0].x_tail)):
        dense_flag |= _lib.BWD_DENSE_GATHER
    _LAST_BWD_FLAGS[0] = dense_flag                      # which path the last backward took (tests)
    if packed is None:
        if x_tails != (0, 2):
            raise NotImplementedError('aladin_amd: the stand-alone backward entry point is the image/caption form')
        geom, pk = align_geometry(Bi, Bc, R, T, D), None
    else:
        geom = packed[0]
        pk = _packed_struct(packed[1], packed[2], packed[3], packed[4] if len(packed) > 4 else None) if have else None
    flags = (_bwd_flags(packed) if have else 0) | dense_flag
    ws = _workspace(lib.aladin_align_bwd_workspace_bytes(C.byref(geom), dense_flag), im.device)
    vi, vs, gi, gs = _set_view(im, im_len_t), _set_view(s, s_len_t), _grad_view(d_im), _grad_view(d_s)
    _lib.check(lib.aladin_align_bwd(C.byref(vi), C.byref(vs), C.byref(geom), C.byref(pk) if pk is not None else None, _ptr(dS), ld_dS,
                                    _ptr(gscale), _ptr(pairs[0] if pairs else None), _ptr(pairs[1] if pairs else None),
                                    C.byref(gi), C.byref(gs), _ptr(ws), flags, _stream()), 'align_bwd')
    return d_im, d_s


_TRIPLET_WS = {}


def _triplet_fused_ok(geom):
    """Shapes aladin_align_triplet_fwd covers (= the fp16 pair kernel's: every training config)."""
    return _pair_kernel_covers(geom) and geom.Bi == geom.Bc and not geom.split and geom.D % 4 == 0 and geom.D <= 1024


def _triplet_forward(im, s, im_len_t, s_len_t, margin, loss_out=None):
    """aladin_align_triplet_fwd: pack + side GEMM + scores + hinge statistics + [pair arg-max | hinge element-wise] in ONE C call.
    -> (loss, S, saved) with saved = (im, s, geom, buf, dS, ws): `buf` is ONE allocation holding xm | xe | y | rnorm, `ws` the node's
    workspace (side scratch, statistics, arg-max table, dS^T) -- both untouched until _triplet_backward.  None when the shape is not
    covered (the caller composes the generic calls)."""
    Bi, R, D = im.shape
    Bc, T, _ = s.shape
    geom = align_geometry(Bi, Bc, R, T, D)
    if not _triplet_fused_ok(geom):
        return None
    lib = _lib.load()
    im, s = _rows_inner_contig(im), _rows_inner_contig(s)
    dev = im.device
    key = (Bi, R, T, D)
    lay = _TRIPLET_WS.get(key)
    if lay is None:
        up = lambda v: (int(v) + 255) // 256 * 256
        o_xe = up(geom.xm_bytes)
        o_y = o_xe + up(max(geom.xe_bytes, 16))
        o_rn = o_y + up(geom.y_bytes)
        lay = (o_xe, o_y, o_rn, o_rn + up(geom.rnorm_bytes), int(lib.aladin_align_triplet_workspace_bytes(C.byref(geom))))
        if len(_TRIPLET_WS) < 256:
            _TRIPLET_WS[key] = lay
    o_xe, o_y, o_rn, n_buf, n_ws = lay
    buf = torch.empty(n_buf, dtype=torch.uint8, device=dev)
    ws = torch.empty(n_ws, dtype=torch.uint8, device=dev)
    S = torch.empty((Bi, Bc), dtype=torch.float32, device=dev)
    dS = torch.empty((Bi, Bc), dtype=torch.float32, device=dev)
    loss = loss_out if loss_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    base = buf.data_ptr()
    pk = _lib.Packed(base, base + o_xe, base + o_y, base + o_rn)
    vi, vs = _set_view(im, im_len_t), _set_view(s, s_len_t)
    _lib.check(lib.aladin_align_triplet_fwd(C.byref(vi), C.byref(vs), C.byref(geom), float(margin), C.byref(pk), _ptr(S), S.stride(0),
                                            _ptr(loss), _ptr(dS), _ptr(ws), _stream()), 'align_triplet_fwd')
    return loss, S, (im, s, geom, buf, dS, ws, (o_xe, o_y, o_rn))


def _packed_from_buf(buf, offs):
    base = buf.data_ptr()
    return _lib.Packed(base, base + offs[0], base + offs[1], base + offs[2])


def _triplet_backward(im, s, im_len_t, s_len_t, geom, pk, dS, ws, gscale, base_workspace=False):
    """aladin_align_triplet_bwd: the row kernel on the table _triplet_forward (or, base_workspace=True, the small-batch heads)
    left in `ws`.  pk: struct aladin_packed of the forward's operands (with rnorm for the fp16 row step)."""
    lib = _lib.load()
    d_im, d_s = _grad_like(im), _grad_like(s)
    flags = _lib.BWD_PARTNERS_FP16 if (_BWD_PARTNERS[0] == 'fp16' and pk.rnorm) else 0
    if flags and _BWD_OWN_ROW_FP16[0]:
        flags |= _lib.BWD_OWN_ROW_FP16
    if base_workspace:
        flags |= _lib.TRIPLET_BWD_BASE_WORKSPACE
    vi, vs, gi, gs = _set_view(im, im_len_t), _set_view(s, s_len_t), _grad_view(d_im), _grad_view(d_s)
    _lib.check(lib.aladin_align_triplet_bwd(C.byref(vi), C.byref(vs), C.byref(geom), C.byref(pk), _ptr(dS), _ptr(gscale), C.byref(gi),
                                            C.byref(gs), _ptr(ws), flags, _stream()), 'align_triplet_bwd')
    return d_im, d_s


class _AlignScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s, im_len_t, s_len_t, x_tail, y_tail):
        need = any(ctx.needs_input_grad[:2])
        if need:
            _check_backward_supported(im, s, x_tail, y_tail)
        S, packed = _align_forward(im, s, im_len_t, s_len_t, x_tail, y_tail, norms=need)
        if need:
            ctx.save_for_backward(im, s, im_len_t, s_len_t, *packed[1:])
            ctx.geom = packed[0]
        return S

    @staticmethod
    def backward(ctx, dS):
        im, s, im_len_t, s_len_t, xm, xe, y, rnorm = ctx.saved_tensors
        dense = False
        if DENSE_BACKWARD and dS.numel() >= DENSE_MIN_PAIRS and not torch.cuda.is_current_stream_capturing():
            probe = _generic_probes.setdefault((tuple(dS.shape), ctx.geom.x_tail, ctx.geom.y_tail), _DensityProbe(unknown=False))
            dense = probe.step(torch.count_nonzero(dS), dS.numel())
        d_im, d_s = _align_backward(im, s, im_len_t, s_len_t, dS, packed=(ctx.geom, xm, xe, y, rnorm), dense=dense)
        return d_im, d_s, None, None, None, None


def _buf_views(buf, geom, offs):
    """(geom, xm, xe, y, rnorm) tensor views of the one-allocation operand buffer of _triplet_forward."""
    xm = buf[:geom.xm_bytes].view(torch.float16)
    xe = buf[offs[0]:offs[0] + max(int(geom.xe_bytes), 16)].view(torch.float16)
    y = buf[offs[1]:offs[1] + geom.y_bytes].view(torch.float16)
    rnorm = buf[offs[2]:offs[2] + geom.rnorm_bytes].view(torch.float32)
    return geom, xm, xe, y, rnorm


class _AlignTriplet(torch.autograd.Function):
    """scores + hinge in one autograd node (reference alad/loss.py:79-159 with return_loss=True):
    dloss/dS never leaves the device-side workspace and the upstream scalar gradient is handed to
    the backward kernels as a device pointer (no element-wise scaling launch).  With the hardest-negative hinge on the
    fp16 pair kernel's shapes -- every training config -- each direction is ONE call into the library
    (aladin_align_triplet_fwd / _bwd; rounds 1-4: six ctypes calls and nine allocations per step)."""

    @staticmethod
    def forward(ctx, im, s, im_len_t, s_len_t, margin, max_violation):
        need = any(ctx.needs_input_grad[:2])
        ctx.fill, _FILL_HINT[0] = _FILL_HINT[0], None
        if need:
            _check_backward_supported(im, s, 0, 2)
        ctx.fused = False
        fused = _triplet_forward(im, s, im_len_t, s_len_t, margin) if need and max_violation else None
        if fused is not None:
            loss, S, (im_c, s_c, geom, buf, dS, ws, offs) = fused
            ctx.save_for_backward(im_c, s_c, im_len_t, s_len_t, buf, dS, ws)
            ctx.geom, ctx.offs, ctx.fused, ctx.pairs, ctx.dense = geom, offs, True, None, False
            ctx.set_materialize_grads(False)
            return loss, S
        S, packed = _align_forward(im, s, im_len_t, s_len_t, norms=need)
        loss, dS, pairs = _hinge_raw(S, margin, max_violation, need, want_pairs=True)
        if need:
            ctx.save_for_backward(im, s, im_len_t, s_len_t, dS, *packed[1:])
            ctx.geom = packed[0]
            ctx.pairs = pairs
        # sum of violations: dloss/dS is dense while most pairs violate the margin (the previous steps' pair counts say)
        ctx.dense = False
        if need and not max_violation:
            ctx.dense = _density_probe.step(pairs[1], S.shape[0] * S.shape[1])
        ctx.set_materialize_grads(False)
        return loss, S

    @staticmethod
    def backward(ctx, g_loss, g_scores):
        if g_loss is None and g_scores is None:
            return None, None, None, None, None, None
        if ctx.fused:
            im, s, im_len_t, s_len_t, buf, dS, ws = ctx.saved_tensors
            if g_scores is None:
                # the argmax table is already there (forward): only the row kernel is left
                d_im, d_s = _triplet_backward(im, s, im_len_t, s_len_t, ctx.geom, _packed_from_buf(buf, ctx.offs), dS, ws,
                                              g_loss.to(torch.float32).contiguous())
                return d_im, d_s, None, None, None, None
            packed = _buf_views(buf, ctx.geom, ctx.offs)
        else:
            im, s, im_len_t, s_len_t, dS, xm, xe, y, rnorm = ctx.saved_tensors
            packed = (ctx.geom, xm, xe, y, rnorm)
        if g_scores is None:
            # the training path: only the loss is differentiated; dloss/dS (<= 3B non-zeros with the hardest-negative
            # hinge) never leaves the device and the upstream scalar goes to the kernels as a pointer
            g = g_loss.to(torch.float32).contiguous()
            d_im, d_s = _align_backward(im, s, im_len_t, s_len_t, dS, gscale=g, packed=packed, pairs=ctx.pairs, dense=ctx.dense, fill=ctx.fill)
        else:
            # the returned score matrix was used too (the reference's S carries grad, alad/loss.py:151-159):
            # total dS = g_loss * dloss/dS + g_scores, generally dense
            total = g_scores.to(torch.float32)
            if g_loss is not None:
                total = total + dS * g_loss.to(torch.float32)
            d_im, d_s = _align_backward(im, s, im_len_t, s_len_t, total.contiguous(), packed=packed, dense=True, fill=ctx.fill)
        return d_im, d_s, None, None, None, None


def _check_sets(im_set, s_seq, im_len, s_len):
    _require_gpu(im_set, s_seq)
    if im_set.dim() != 3 or s_seq.dim() != 3:
        raise ValueError('aladin_amd: im_set (B,R,D) and s_seq (B,T,D) expected')
    if len(im_len) != im_set.shape[0] or len(s_len) != s_seq.shape[0]:
        raise ValueError('aladin_amd: one length per sample expected')
    return lengths_tensor(im_len, im_set.device), lengths_tensor(s_len, im_set.device)


def alignment_triplet_loss(im_set, s_seq, im_len, s_len, margin, max_violation):
    """(loss, S) of AlignmentContrastiveLoss(aggregation='MrSw') in one fused autograd node.  Both outputs
    are differentiable (as in the reference, alad/loss.py:151-159); when only the loss is back-propagated --
    every shipped config: S is fed, detached, to the distillation loss, alad/loss.py:370 -- the backward
    stays on the sparse dloss/dS path."""
    im_len_t, s_len_t = _check_sets(im_set, s_seq, im_len, s_len)
    if im_set.shape[0] != s_seq.shape[0]:
        raise ValueError('aladin_amd: the contrastive loss needs a square score matrix, got (%d, %d) '
                         '(the reference fails in diag/expand_as, alad/loss.py:43-45)' % (im_set.shape[0], s_seq.shape[0]))
    if torch.is_grad_enabled() and (im_set.requires_grad or s_seq.requires_grad):
        im_set, s_seq = _pad_features(im_set, s_seq)
    _FILL_HINT[0] = None if max_violation else _caption_fill(s_len, s_seq.shape[1])
    return _AlignTriplet.apply(im_set, s_seq, im_len_t, s_len_t, margin, max_violation)


def _host_lengths(lens):
    return [int(v) for v in (lens.tolist() if isinstance(lens, torch.Tensor) else lens)]


# ------------------------------------------------------------------------------------------------
# Length-bucketed evaluation grids.  The packed geometry pads the max side to its region classes (32 rows, 32 + up to 8
# side rows, 48, 48 + up to 8 side rows, 64, 96: align_fwd.hip geometry) and the sum side to 8, 16, 24, 32, 40, 48, 64 or 96
# words: one launch over the whole grid pays
# for the LONGEST image and caption at every pair.  Real sets are ragged (COCO: 10-50 boxes, captions of ~12 tokens),
# so the samples of each side are grouped by the tile class their own length needs, every (image class x caption
# class) block is scored with its own geometry, and the blocks are laid back in the callers' order.  A score only
# depends on its own image and caption rows (masked positions are zero rows whatever the class), so the scores are
# those of the single launch up to the summation order of a different kernel variant (~1e-7 in split precision).
# ------------------------------------------------------------------------------------------------
X_CLASS_BOUNDS = (32, 40, 48, 56, 64, 96)  # scored max-side positions (incl. the one masked position kept for the zero fill)
Y_CLASS_BOUNDS = (8, 16, 24, 32, 40, 48, 64, 96)   # scored sum-side positions (8 / 24 / 40: two captions share one / three / five 16-word tiles)
BUCKET_MIN_PAIRS = 1 << 18                 # below this a grid is one launch
BUCKET_MIN_SAMPLES = 64                    # smaller classes join the next longer one
BUCKET_MIN_GAIN = 0.10                     # padded work saved before bucketing is worth its extra launches


def _needed_positions(lens, tail, total, keep_masked):
    """Positions of the packed operand sample k needs: its scored positions (len - 1 - tail, clamped to the set), plus one
    masked position on the max side when the sample is shorter than the set (alad/loss.py:116,124: the zero fill takes
    part in the max)."""
    cap = max(total - 1 - tail, 1)
    out = []
    for v in lens:
        n = min(max(int(v) - 1 - tail, 1), cap)
        out.append(min(n + 1, cap) if keep_masked and n < cap else n)
    return out


def bucket_plan(x_need, y_need):
    """-> (x_groups, y_groups) lists of index lists (callers' order inside a group), or None when one launch is the better
    choice.  x_need / y_need: positions per sample (_needed_positions)."""
    def groups(need, bounds):
        cls = [[] for _ in bounds]
        for k, n in enumerate(need):
            for c, b in enumerate(bounds):
                if n <= b or c == len(bounds) - 1:
                    cls[c].append(k)
                    break
        for c in range(len(cls) - 1):                               # small classes join the next longer one
            if 0 < len(cls[c]) < BUCKET_MIN_SAMPLES:
                cls[c + 1] = sorted(cls[c] + cls[c + 1])
                cls[c] = []
        last = [c for c in range(len(cls)) if cls[c]]
        if len(last) > 1 and len(cls[last[-1]]) < BUCKET_MIN_SAMPLES:   # a small LAST class takes its neighbour in
            cls[last[-1]] = sorted(cls[last[-2]] + cls[last[-1]])
            cls[last[-2]] = []
        return [(g, bounds[c]) for c, g in enumerate(cls) if g]
    if len(x_need) * len(y_need) < BUCKET_MIN_PAIRS:
        return None
    gx, gy = groups(x_need, X_CLASS_BOUNDS), groups(y_need, Y_CLASS_BOUNDS)
    if len(gx) == 1 and len(gy) == 1:
        return None

    def padded(n, bounds):
        return next((b for b in bounds if n <= b), bounds[-1])
    one = len(x_need) * len(y_need) * padded(max(x_need), X_CLASS_BOUNDS) * padded(max(y_need), Y_CLASS_BOUNDS)
    work = sum(len(a) * len(b) * padded(max(x_need[k] for k in a), X_CLASS_BOUNDS) * padded(max(y_need[k] for k in b), Y_CLASS_BOUNDS)
               for a, _ in gx for b, _ in gy)
    if work > (1.0 - BUCKET_MIN_GAIN) * one:
        return None
    return [a for a, _ in gx], [b for b, _ in gy]


def _index_tensor(ids, device, dtype=torch.int64):
    """Index list -> device tensor WITHOUT making the host wait for the device (torch.tensor(..., device=...) is a blocking
    copy: it drains the stream, and a grid scored in blocks would serialise host and GPU work block by block)."""
    return torch.tensor(ids, dtype=dtype).to(device, non_blocking=True)


class GridPlan:
    """The length classes of one evaluation grid, with everything the blocks need already on the device: per-class index
    tensors and the permutation that lays the class-ordered blocks back in the callers' order.  Built once per
    (lengths, device) and cached: validation scores the same sets twice per epoch (i2t, t2i) and every epoch again."""

    def __init__(self, gx, gy, device):
        self.gx, self.gy = gx, gy
        self.ix = [_index_tensor(g, device) for g in gx]
        self.iy = [_index_tensor(g, device) for g in gy]
        Bx, By = sum(len(g) for g in gx), sum(len(g) for g in gy)
        inv_x = torch.empty(Bx, dtype=torch.int64)
        inv_x[torch.tensor([k for g in gx for k in g], dtype=torch.int64)] = torch.arange(Bx, dtype=torch.int64)
        inv_y = torch.empty(By, dtype=torch.int64)
        inv_y[torch.tensor([k for g in gy for k in g], dtype=torch.int64)] = torch.arange(By, dtype=torch.int64)
        self.inv_x, self.inv_y = inv_x.to(device, non_blocking=True), inv_y.to(device, non_blocking=True)
        self.extra = {}                                   # per-caller attachments (store views)

    def assemble(self, blocks):
        """Blocks scored in class order -> the (Bx, By) matrix in the callers' order."""
        rows = [torch.cat([blocks[(a, b)] for b in range(len(self.gy))], dim=1) for a in range(len(self.gx))]
        return torch.cat(rows, dim=0).index_select(0, self.inv_x).index_select(1, self.inv_y)


_PLAN_CACHE = {}


def grid_plan(key, x_need_fn, y_need_fn, device):
    """Cached GridPlan (or None: one launch) for `key`; the *_need_fn callables are only evaluated on a miss."""
    key = key + (str(device),)
    if key in _PLAN_CACHE:
        return _PLAN_CACHE[key]
    plan = bucket_plan(x_need_fn(), y_need_fn())
    entry = GridPlan(plan[0], plan[1], device) if plan is not None else None
    if len(_PLAN_CACHE) >= 8:
        _PLAN_CACHE.clear()
    _PLAN_CACHE[key] = entry
    return entry


def _scores_nograd(xs, ys, x_len, y_len, x_tail, y_tail, precision):
    """(Bx, By) scores of a max-side set xs (Bx, N, D) against a sum-side set ys (By, M, D) outside autograd:
    the evaluation path.  Four things the differentiable path does not do:
      * operands in the evaluation precision (split fp16 by default: rank-exact Recall);
      * both sets are trimmed to the positions that can matter.  encode_data pads every set to 71 positions
        (alad/evaluation.py:98-99); positions past a set's length are masked to 0 by alad/loss.py:103-116
        whatever they hold.  On the SUM side a masked position adds exactly 0, so the set is cut at the
        longest length.  On the MAX side the masked positions still take part in the max as zeros
        (`max(real dots, 0)`, alad/loss.py:116,124) for every sample shorter than the padded set, so ONE
        masked position is kept: the set is cut at the longest length + 1 (or not at all when some sample
        fills it).  The 70 x 68 padded block per pair shrinks to the real one and no score changes;
      * large ragged grids are scored in length classes (bucket_plan above): a pair pays for the tile class of ITS image
        and caption, not for the longest of the evaluation set;
      * the sum side is chunked so the side-row scratch of the score kernel stays under E_SCRATCH_LIMIT
        (16 GB at 5000 x 25000 otherwise); a score does not depend on the chunking."""
    x_len, y_len = _host_lengths(x_len), _host_lengths(y_len)
    dev = xs.device
    plan = None
    if len(x_len) * len(y_len) >= BUCKET_MIN_PAIRS:
        plan = grid_plan(('dense', tuple(x_len), tuple(y_len), x_tail, y_tail, xs.shape[1], ys.shape[1]),
                         lambda: _needed_positions(x_len, x_tail, xs.shape[1], True),
                         lambda: _needed_positions(y_len, y_tail, ys.shape[1], False), dev)
    if plan is None:
        return _scores_nograd_block(xs, ys, x_len, y_len, x_tail, y_tail, precision)
    blocks = {}
    ysub = []
    for b, ib in zip(plan.gy, plan.iy):
        m_eff = min(ys.shape[1], max(2 + y_tail, max(y_len[k] for k in b)))
        ysub.append((ys[:, :m_eff].index_select(0, ib), [y_len[k] for k in b]))
    for ia, (a, ixa) in enumerate(zip(plan.gx, plan.ix)):
        n_eff = min(xs.shape[1], max(2 + x_tail, max(x_len[k] for k in a) + 1))
        xa = xs[:, :n_eff].index_select(0, ixa)
        la = [x_len[k] for k in a]
        for ib, (yb, lb) in enumerate(ysub):
            # the block keeps the FULL set's zero-fill rule: a sample is "shorter than the padded set" relative to xs, not to
            # its class, which _scores_nograd_block reproduces because every class member is cut at its class's longest + 1
            blocks[(ia, ib)] = _scores_nograd_block(xa, yb, la, lb, x_tail, y_tail, precision, x_total=xs.shape[1])
    return plan.assemble(blocks)


def _scores_nograd_block(xs, ys, x_len, y_len, x_tail, y_tail, precision, x_total=None):
    """One geometry for the whole block (see _scores_nograd).  x_total: positions of the set the block's max side was cut
    from (a sample that fills IT has no masked position; default: xs itself)."""
    x_total = xs.shape[1] if x_total is None else x_total
    longest = max(x_len)
    n_eff = min(xs.shape[1], max(2 + x_tail, longest + (1 if longest < x_total else 0)))
    m_eff = min(ys.shape[1], max(2 + y_tail, max(y_len)))
    xs, ys = xs[:, :n_eff], ys[:, :m_eff]
    dev = xs.device
    x_len_t, y_len_t = lengths_tensor(x_len, dev), lengths_tensor(y_len, dev)
    Bx, By, D = xs.shape[0], ys.shape[0], xs.shape[2]
    geom = align_geometry(Bx, By, n_eff, m_eff, D, x_tail, y_tail, precision)
    if geom.e_bytes <= E_SCRATCH_LIMIT:
        return _align_forward(xs, ys, x_len_t, y_len_t, x_tail, y_tail, precision, norms=False)[0]
    step = max(geom.cap_unit, int(By * E_SCRATCH_LIMIT // geom.e_bytes) // geom.cap_unit * geom.cap_unit)
    xm, xe = pack_images(xs, x_len_t, geom)
    S = torch.empty((Bx, By), dtype=torch.float32, device=dev)
    for j0 in range(0, By, step):
        j1 = min(By, j0 + step)
        g = align_geometry(Bx, j1 - j0, n_eff, m_eff, D, x_tail, y_tail, precision)        # same max-side layout
        y = pack_captions(ys[j0:j1], y_len_t[j0:j1].contiguous(), g)
        scores_from_packed(xm, xe, y, g, out=S[:, j0:j1])
    return S


def alignment_scores(im_set, s_seq, im_len, s_len, aggregation='MrSw', precision=None):
    """S (Bi, Bc); replaces reference alad/loss.py:80-135.  Differentiable when autograd is on and an
    input requires grad (fp16 operands); otherwise the evaluation path of _scores_nograd in the evaluation
    precision (`precision` overrides set_eval_precision()).
      'MrSw'  sum over words of the max over regions                       (:124-125)
      'MwSr'  sum over regions of the max over words: the same kernels with the two sets swapped --
              captions on the max side (tail 2), images on the sum side (tail 0) -- transposed (:134-135)
      'symm'  MrSw + MwSr                                                  (:130-133)"""
    if aggregation not in ('MrSw', 'MwSr', 'symm'):
        raise NotImplementedError('aladin_amd: aggregation %r' % (aggregation,))
    if not (torch.is_grad_enabled() and (im_set.requires_grad or s_seq.requires_grad)):
        _require_gpu(im_set, s_seq)
        if im_set.dim() != 3 or s_seq.dim() != 3:
            raise ValueError('aladin_amd: im_set (B,R,D) and s_seq (B,T,D) expected')
        if len(im_len) != im_set.shape[0] or len(s_len) != s_seq.shape[0]:
            raise ValueError('aladin_amd: one length per sample expected')
        if im_set.shape[2] != s_seq.shape[2]:
            raise ValueError('aladin_amd: feature sizes differ (%d vs %d)' % (im_set.shape[2], s_seq.shape[2]))
        prec = precision if precision is not None else _EVAL_PRECISION[0]
        with torch.no_grad():
            S = None
            if aggregation in ('MrSw', 'symm'):
                S = _scores_nograd(im_set, s_seq, im_len, s_len, 0, 2, prec)
            if aggregation in ('MwSr', 'symm'):
                St = _scores_nograd(s_seq, im_set, s_len, im_len, 2, 0, prec).t()
                S = St if S is None else S + St
        return S
    if precision not in (None, 'fp16'):
        raise ValueError('aladin_amd: differentiable alignment scores use fp16 operands (split precision is forward-only)')
    im_len_t, s_len_t = _check_sets(im_set, s_seq, im_len, s_len)
    im_set, s_seq = _pad_features(im_set, s_seq)
    if aggregation == 'MrSw':
        return _AlignScores.apply(im_set, s_seq, im_len_t, s_len_t, 0, 2)
    if aggregation == 'MwSr':
        return _AlignScores.apply(s_seq, im_set, s_len_t, im_len_t, 2, 0).t()
    if aggregation == 'symm':
        return _AlignScores.apply(im_set, s_seq, im_len_t, s_len_t, 0, 2) + \
            _AlignScores.apply(s_seq, im_set, s_len_t, im_len_t, 2, 0).t()


class _ScanScores(torch.autograd.Function):
    """aggregation='scan-sentences' (reference alad/loss.py:136-149)."""

    @staticmethod
    def forward(ctx, im, s, im_len_t, s_len_t):
        lib = _lib.load()
        im = _rows_inner_contig(im)
        s = _rows_inner_contig(s)
        Bi, R, D = im.shape
        Bc, T, _ = s.shape
        S = torch.empty((Bi, Bc), dtype=torch.float32, device=im.device)
        ws = _workspace(lib.aladin_scan_workspace_bytes(Bi, Bc, R, T, D, 0), im.device)
        _lib.check(lib.aladin_scan_fwd(_ptr(im), im.stride(0), im.stride(1), _ptr(im_len_t), _ptr(s), s.stride(0), s.stride(1),
                                       _ptr(s_len_t), Bi, Bc, R, T, D, _ptr(S), S.stride(0), _ptr(ws), _stream()), 'scan_fwd')
        ctx.save_for_backward(im, s, im_len_t, s_len_t)
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.load()
        im, s, im_len_t, s_len_t = ctx.saved_tensors
        Bi, R, D = im.shape
        Bc, T, _ = s.shape
        dS = dS.contiguous()
        d_im = torch.empty((Bi, R, D), dtype=torch.float32, device=im.device)
        d_s = torch.empty((Bc, T, D), dtype=torch.float32, device=im.device)
        ws = _workspace(lib.aladin_scan_workspace_bytes(Bi, Bc, R, T, D, 1), im.device)
        _lib.check(lib.aladin_scan_bwd(_ptr(im), im.stride(0), im.stride(1), _ptr(im_len_t), _ptr(s), s.stride(0), s.stride(1),
                                       _ptr(s_len_t), Bi, Bc, R, T, D, _ptr(dS), dS.shape[1], _ptr(None), _ptr(d_im), _ptr(d_s),
                                       _ptr(ws), _stream()), 'scan_bwd')
        return d_im, d_s, None, None


def alignment_scan_scores(im_set, s_seq, im_len, s_len):
    """(Bi, Bc) 'scan-sentences' scores, differentiable; replaces reference alad/loss.py:136-149 (+ :80-116)."""
    im_len_t, s_len_t = _check_sets(im_set, s_seq, im_len, s_len)
    if s_seq.shape[1] < 4:
        raise ValueError('aladin_amd: captions need at least 4 positions (token 0 and the last two are dropped)')
    return _ScanScores.apply(im_set, s_seq, im_len_t, s_len_t)


class _NormSum(torch.autograd.Function):
    """(B,N,D) set -> (B,D) sum of its L2-normalised rows 1 .. len-1-tail."""

    @staticmethod
    def forward(ctx, x, len_t, tail):
        x = _rows_inner_contig(x)
        B, N, D = x.shape
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().aladin_normsum_fwd(_ptr(x), x.stride(0), x.stride(1), _ptr(len_t), B, N, D, tail, _ptr(out),
                                                  _stream()), 'normsum_fwd')
        ctx.save_for_backward(x, len_t)
        ctx.tail = tail
        return out

    @staticmethod
    def backward(ctx, g):
        x, len_t = ctx.saved_tensors
        B, N, D = x.shape
        g = g.contiguous()
        d_x = torch.empty((B, N, D), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().aladin_normsum_bwd(_ptr(x), x.stride(0), x.stride(1), _ptr(len_t), B, N, D, ctx.tail,
                                                  _ptr(g), _ptr(d_x), _stream()), 'normsum_bwd')
        return d_x, None, None


def alignment_sum_scores(im_set, s_seq, im_len, s_len, mean=False):
    """'sum' / 'mean' pooling (reference alad/loss.py:120-123): the double sum of masked cosines equals
    the dot product of the summed unit vectors, so no region x word tensor is formed."""
    im_len_t, s_len_t = _check_sets(im_set, s_seq, im_len, s_len)
    S = _DotScores.apply(_NormSum.apply(im_set, im_len_t, 0), _NormSum.apply(s_seq, s_len_t, 2))
    if mean:
        S = S / float((im_set.shape[1] - 1) * (s_seq.shape[1] - 3))
    return S


# ------------------------------------------------------------------------------------------------
# One namespace for callers (`from aladin_amd import ops; ops.hinge_loss(...)`): the losses, the loss heads and the retrieval
# functions live in ops_losses / ops_heads / ops_retrieval (round 5 split of a 1400-line module) and are looked up lazily, so that
# ops_heads may import THIS module (it builds on the alignment nodes) without a cycle.
# ------------------------------------------------------------------------------------------------
_SUBMODULES = ('ops_losses', 'ops_heads', 'ops_retrieval')


def __getattr__(name):
    import importlib
    if name.startswith('__'):
        raise AttributeError(name)
    for m in _SUBMODULES:
        mod = importlib.import_module('.' + m, __package__)
        if name in mod.__dict__:
            return mod.__dict__[name]
    raise AttributeError('module %r has no attribute %r' % (__name__, name))

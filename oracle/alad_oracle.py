"""CPU oracle for ALADIN's alignment-scoring / matching-retrieval hot path.  TEST INFRASTRUCTURE.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path (``aladin_amd``) runs the
HIP kernels in ``aladin_amd/csrc`` and raises if the extension is missing.

It is a numpy restatement (closed forms) of the reference's algorithm; each
function cites the reference lines (relative to /root/reference) it follows.  The reference is pure
Python/PyTorch and holds no tests or golden vectors of its own, so parity is PINNED against outputs
of the reference itself: ``tests/golden/make_golden.py`` imports ``alad.loss`` /
``alad.recall_auxiliary`` / ``alad.evaluation`` / ``alad.alad_model`` from /root/reference (in the
build container only) and commits their outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those fixtures.
"""
import numpy as np

F_EPS = 1e-12          # torch.nn.functional.normalize default eps (alad/loss.py:80-81)


# ----------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------
def normalize_rows(x, eps=F_EPS):
    """x / max(||x||_2, eps) along the last axis -- F.normalize(p=2, dim=2), alad/loss.py:80-81."""
    n = np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=-1, keepdims=True))
    return (x / np.maximum(n, eps)).astype(x.dtype)


def l2norm(x):
    """X / sqrt(sum_dim1 X^2), NO eps (zero rows -> NaN) -- alad/utils.py:134-139."""
    with np.errstate(invalid='ignore', divide='ignore'):
        n = np.sqrt(np.sum(x * x, axis=1, keepdims=True))
        return x / n


# ----------------------------------------------------------------------------------------------
# alignment scores (alad/loss.py:79-135)
# ----------------------------------------------------------------------------------------------
def masked_alignments(im_set, s_seq, im_len, s_len, dtype=np.float32):
    """The (Bi, Bc, R', T') tensor of alad/loss.py:97-116: region x word cosines after dropping
    region 0 / token 0 / the last two token columns (:87-88), zero where r >= im_len-1 or
    w >= s_len-3 (:89-90,103-116).  Materialises Bi*Bc*R'*T' floats: small cases only."""
    im = normalize_rows(np.asarray(im_set, dtype))[:, 1:, :]
    s = normalize_rows(np.asarray(s_seq, dtype))[:, 1:-2, :]
    Bi, Rp, D = im.shape
    Bc, Tp, _ = s.shape
    A = (im.reshape(Bi * Rp, D) @ s.reshape(Bc * Tp, D).T).reshape(Bi, Rp, Bc, Tp)
    A = np.ascontiguousarray(A.transpose(0, 2, 1, 3))
    Li = np.asarray(im_len) - 1
    Lj = np.asarray(s_len) - 3
    rmask = np.arange(Rp)[None, :] >= Li[:, None]          # (Bi, R')
    wmask = np.arange(Tp)[None, :] >= Lj[:, None]          # (Bc, T')
    A[np.broadcast_to(rmask[:, None, :, None] | wmask[None, :, None, :], A.shape)] = 0
    return A


def alignment_scores(im_set, s_seq, im_len, s_len, aggregation='MrSw', dtype=np.float32,
                     block=32):
    """S (Bi, Bc) for the pooling modes of alad/loss.py:120-135.  Blocked over images so that
    B=256 needs ~50 MB instead of the reference's 16 GB of expanded operands."""
    im_set = np.asarray(im_set)
    Bi = im_set.shape[0]
    out = []
    for i0 in range(0, Bi, block):
        A = masked_alignments(im_set[i0:i0 + block], s_seq, im_len[i0:i0 + block], s_len, dtype)
        if aggregation == 'MrSw':                       # :124-125
            S = A.max(2).sum(2)
        elif aggregation == 'MrAVGw':                   # :126-129
            S = A.max(2).sum(2) / (np.asarray(s_len, dtype) - 3)[None, :]
        elif aggregation == 'MwSr':                     # :134-135
            S = A.max(3).sum(2)
        elif aggregation == 'symm':                     # :130-133
            S = A.max(2).sum(2) + A.max(3).sum(2)
        elif aggregation == 'sum':                      # :120-121
            S = A.sum((2, 3))
        elif aggregation == 'mean':                     # :122-123 (divides by the padded sizes)
            S = A.mean((2, 3))
        else:
            raise ValueError('aggregation %r not restated' % (aggregation,))
        out.append(S)
    return np.concatenate(out, 0).astype(dtype)


# ----------------------------------------------------------------------------------------------
# hinge / VSE++ (alad/loss.py:42-67)
# ----------------------------------------------------------------------------------------------
def hinge_loss(S, margin, max_violation, return_grad=False):
    """loss (and dloss/dS) of Contrastive.compute_contrastive_loss, alad/loss.py:42-67."""
    S = np.asarray(S)
    if S.shape[0] != S.shape[1]:
        raise ValueError('hinge loss needs a square score matrix')
    B = S.shape[0]
    d = np.diag(S)
    cs = np.clip(margin + S - d[:, None], 0, None)        # :49   compare with the row's diagonal
    ci = np.clip(margin + S - d[None, :], 0, None)        # :52   compare with the column's diagonal
    np.fill_diagonal(cs, 0)
    np.fill_diagonal(ci, 0)                               # :55-60
    dS = np.zeros_like(S)
    if max_violation:                                     # :63-65
        loss = cs.max(1).sum() + ci.max(0).sum()
        if return_grad:
            js = cs.argmax(1)
            for i in range(B):
                if cs[i, js[i]] > 0:
                    dS[i, js[i]] += 1
                    dS[i, i] -= 1
            is_ = ci.argmax(0)
            for j in range(B):
                if ci[is_[j], j] > 0:
                    dS[is_[j], j] += 1
                    dS[j, j] -= 1
    else:
        loss = cs.sum() + ci.sum()                        # :67
        if return_grad:
            P = (cs > 0).astype(S.dtype)
            Q = (ci > 0).astype(S.dtype)
            dS = P + Q - np.diag(P.sum(1)) - np.diag(Q.sum(0))
    loss = S.dtype.type(loss)
    return (loss, dS) if return_grad else loss


# ----------------------------------------------------------------------------------------------
# backward of S w.r.t. the raw sets (autograd of alad/loss.py:80-125), MrSw only
# ----------------------------------------------------------------------------------------------
def alignment_scores_backward(im_set, s_seq, im_len, s_len, dS, dtype=np.float64):
    """(d im_set, d s_seq) for upstream gradient dS (Bi, Bc).  Python loops: small cases only."""
    im_set = np.asarray(im_set, dtype)
    s_seq = np.asarray(s_seq, dtype)
    Bi, R, D = im_set.shape
    Bc, T, _ = s_seq.shape
    Rp, Tp = R - 1, T - 3
    n_im = np.maximum(np.sqrt((im_set ** 2).sum(-1, keepdims=True)), F_EPS)
    n_s = np.maximum(np.sqrt((s_seq ** 2).sum(-1, keepdims=True)), F_EPS)
    ih, sh = im_set / n_im, s_seq / n_s
    dih, dsh = np.zeros_like(ih), np.zeros_like(sh)
    for i in range(Bi):
        Li = im_len[i] - 1
        for j in range(Bc):
            g = dS[i, j]
            if g == 0:
                continue
            Lj = s_len[j] - 3
            if Li <= 0 or Lj <= 0:                                 # everything masked: no gradient
                continue
            A = ih[i, 1:1 + Li] @ sh[j, 1:1 + Lj].T            # (Li, Lj)
            rstar = A.argmax(0)
            for w in range(Lj):
                r = rstar[w]
                if Li < Rp and A[r, w] <= 0:                   # the zero fill of :116 won the max
                    continue
                dih[i, 1 + r] += g * sh[j, 1 + w]
                dsh[j, 1 + w] += g * ih[i, 1 + r]
    dim = (dih - ih * (ih * dih).sum(-1, keepdims=True)) / n_im      # normalize backward
    ds = (dsh - sh * (sh * dsh).sum(-1, keepdims=True)) / n_s
    return dim, ds


# ----------------------------------------------------------------------------------------------
# matching head (alad/loss.py:8-11,179-186) and listnet distillation (alad/loss.py:369-370,427-447)
# ----------------------------------------------------------------------------------------------
def dot_scores(im, s):
    """im @ s.T -- dot_sim, alad/loss.py:8-11 (also recall_auxiliary.py:30, evaluation.py:196)."""
    return np.asarray(im) @ np.asarray(s).T


def _softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def listnet_loss(teacher, student, return_grad=False, temperature=6.0, eps=1e-10):
    """DistillationLoss(mode='listnet'), alad/loss.py:427-445.  Gradient w.r.t. the student only
    (the teacher is detached at :370)."""
    T = np.asarray(teacher)
    M = np.asarray(student)
    B = M.shape[0]
    loss = 0.0
    dM = np.zeros(M.shape, np.float64)
    for axis in (1, 0):                                        # :431-436 rows, :438-443 columns
        Q = _softmax(M.astype(np.float64) * temperature, axis)
        P = _softmax(T.astype(np.float64), axis)
        loss += np.mean(-np.sum(P * np.log(Q + eps), axis=axis))
        if return_grad:
            W = P * Q / (Q + eps)
            n = M.shape[axis]
            dM += temperature / (M.shape[1 - axis]) * (Q * W.sum(axis=axis, keepdims=True) - W)
    loss = M.dtype.type(loss)
    return (loss, dM.astype(M.dtype)) if return_grad else loss


def scan_sentences_scores(im_set, s_seq, im_len, s_len, dS=None):
    """aggregation='scan-sentences', alad/loss.py:136-149, pair by pair in float64.

    Per pair, with A the masked cosine block (:99-116): N = relu(A) L2-normalised over REGIONS per word
    (:137-138), W = softmax over the valid WORDS of each valid region row (:139-140), the attended
    sentence vector att_r = sum_w W[r,w] s_w (:142-145) and S = sum_{r<Li} cos(i_r, att_r) (:146-149),
    where <i_r, att_r> = sum_w W[r,w] A[r,w] and |att_r|^2 = W_r G W_r^T with G the caption's Gram matrix.
    A pair whose caption has no scored word is NaN (softmax over an empty row), as in the reference.

    With dS (Bi, Bc) also returns (d im_set, d s_seq) of sum(dS * S): the analytic gradient of the
    masked expression.  (The reference's own autograd returns NaN as soon as any image is shorter than
    the batch maximum -- 0 * NaN from its -inf rows -- so it is only comparable on full-length batches.)"""
    im = np.asarray(im_set, np.float64)
    s = np.asarray(s_seq, np.float64)
    Bi, Bc = im.shape[0], s.shape[0]
    ni = np.sqrt((im * im).sum(-1, keepdims=True))
    ns = np.sqrt((s * s).sum(-1, keepdims=True))
    xn = (im / np.maximum(ni, F_EPS))[:, 1:, :]
    yn = (s / np.maximum(ns, F_EPS))[:, 1:-2, :]
    S = np.zeros((Bi, Bc), np.float64)
    dxn = np.zeros_like(xn)
    dyn = np.zeros_like(yn)
    for i in range(Bi):
        Li = max(int(im_len[i]) - 1, 0)
        for j in range(Bc):
            Lj = max(int(s_len[j]) - 3, 0)
            if Li == 0:
                continue
            if Lj == 0:
                S[i, j] = np.nan
                continue
            x, y = xn[i, :Li], yn[j, :Lj]
            A = x @ y.T
            P = np.maximum(A, 0.0)
            c = np.maximum(np.sqrt((P * P).sum(0)), 1e-12)
            N = P / c
            E = np.exp(N - N.max(1, keepdims=True))
            W = E / E.sum(1, keepdims=True)
            G = y @ y.T
            u = (W * A).sum(1)
            t = W @ G
            q = (W * t).sum(1)
            n = np.sqrt(q)
            a = np.maximum(np.sqrt((x * x).sum(1)), 1e-8)
            m = np.maximum(n, 1e-8)
            cos = u / (a * m)
            S[i, j] = cos.sum()
            if dS is None or dS[i, j] == 0:
                continue
            g = float(dS[i, j])
            k = np.where(n > 1e-8, cos / m ** 2, 0.0)              # d cos / d q = -k / 2 (0 when |att| is clamped)
            dW = g * (A / (a * m)[:, None] - k[:, None] * t)
            dZ = W * (dW - (W * dW).sum(1, keepdims=True))
            dot = (N * dZ).sum(0)
            dP = (dZ - N * dot) / c
            dA = g * W / (a * m)[:, None] + dP * (A > 0)
            H = g * (W * k[:, None]).T @ W
            dxn[i, :Li] += dA @ y
            dyn[j, :Lj] += dA.T @ x - H @ y
    S = S.astype(np.float32)
    if dS is None:
        return S

    def norm_bwd(vn, dvn, nrm):
        return (dvn - vn * (vn * dvn).sum(-1, keepdims=True)) / np.maximum(nrm, F_EPS)

    d_im = np.zeros_like(im)
    d_s = np.zeros_like(s)
    d_im[:, 1:, :] = norm_bwd(xn, dxn, ni[:, 1:, :])
    d_s[:, 1:-2, :] = norm_bwd(yn, dyn, ns[:, 1:-2, :])
    return S, d_im.astype(np.float32), d_s.astype(np.float32)


def distill_mse(teacher, student, wb, return_grad=False):
    """DistillationLoss(mode='mse'), alad/loss.py:371-373: mean((student*wb[0] + wb[1] - teacher)^2);
    gradients w.r.t. the student and the learnable pair wb (:366)."""
    T = np.asarray(teacher, np.float64)
    M = np.asarray(student, np.float64)
    w0, w1 = float(wb[0]), float(wb[1])
    r = M * w0 + w1 - T
    loss = np.float32(np.mean(r * r))
    if not return_grad:
        return loss
    n = r.size
    return loss, (2.0 * r * w0 / n).astype(np.float32), np.array([np.sum(2.0 * r * M) / n, np.sum(2.0 * r) / n], np.float32)


def distill_contrastive(teacher, student, margin=0.2, return_grad=False):
    """DistillationLoss(mode='contrastive'), alad/loss.py:401-425, as written: the teacher's diagonal
    is set to 0 (not -inf, :405), its row / column argmax picks whole COLUMNS of cost_s (:417-418) and
    whole ROWS of cost_im (:420-421), and neither cost has its diagonal cleared.  So
        loss = sum_k sum_i relu(m + S[i, a_k] - S[i, i]) + sum_k sum_j relu(m + S[b_k, j] - S[j, j]),
    a_k = argmax_j T0[k, j], b_k = argmax_i T0[i, k]."""
    T0 = np.array(teacher, np.float64)
    M = np.asarray(student, np.float64)
    B = M.shape[0]
    T0[np.arange(B), np.arange(B)] = 0.0
    col_count = np.bincount(np.argmax(T0, axis=1), minlength=B).astype(np.float64)   # c_j
    row_count = np.bincount(np.argmax(T0, axis=0), minlength=B).astype(np.float64)   # r_i
    diag = np.diag(M)
    a = margin + M - diag[:, None]
    b = margin + M - diag[None, :]
    loss = np.float32(np.sum(col_count[None, :] * np.maximum(a, 0)) + np.sum(row_count[:, None] * np.maximum(b, 0)))
    if not return_grad:
        return loss
    ga = col_count[None, :] * (a > 0)
    gb = row_count[:, None] * (b > 0)
    dM = ga + gb
    dM[np.arange(B), np.arange(B)] -= ga.sum(axis=1) + gb.sum(axis=0)
    return loss, dM.astype(np.float32)


def distill_ordinal(teacher, student, margin=0.2, threshold=0.1, stride=3, return_grad=False):
    """DistillationLoss(mode='ordinal'), alad/loss.py:374-399: sort every teacher row (then column)
    ascending, read the student in that order, and ask student[p] + margin <= student[p + stride]
    wherever teacher_sorted[p + stride] >= threshold; mean hinge over those positions, rows + columns.
    An empty selection gives a NaN loss (torch's mean of an empty tensor) and a zero gradient."""
    T = np.asarray(teacher, np.float64)
    M = np.asarray(student, np.float64)
    loss = 0.0
    dM = np.zeros(M.shape, np.float64)
    for axis in (1, 0):
        Tt, Mt = (T, M) if axis == 1 else (T.T, M.T)
        order = np.argsort(Tt, axis=1, kind='stable')
        ts = np.take_along_axis(Tt, order, axis=1)
        so = np.take_along_axis(Mt, order, axis=1)
        diff = so[:, :-stride] - so[:, stride:]
        valid = ts[:, stride:] >= threshold
        n = int(valid.sum())
        with np.errstate(invalid='ignore', divide='ignore'):
            loss += np.sum(np.maximum(margin + diff, 0) * valid) / n if n else np.nan
        if return_grad:
            act = (valid & (margin + diff > 0)).astype(np.float64) / max(n, 1)   # empty selection: NaN loss, zero gradient
            g_sorted = np.zeros(Mt.shape, np.float64)
            g_sorted[:, :-stride] += act
            g_sorted[:, stride:] -= act
            g = np.zeros(Mt.shape, np.float64)
            np.put_along_axis(g, order, g_sorted, axis=1)
            dM += g if axis == 1 else g.T
    loss = np.float32(loss)
    return (loss, dM.astype(np.float32)) if return_grad else loss


def order_scores(im, s):
    """order_sim, alad/loss.py:20-26: score[i, j] = -|| max(s_j - im_i, 0) ||_2."""
    c = np.maximum(np.asarray(s, np.float64)[None, :, :] - np.asarray(im, np.float64)[:, None, :], 0.0)
    return (-np.sqrt(np.sum(c * c, axis=2))).astype(np.float32)


def order_scores_backward(im, s, G):
    """(d im, d s) of sum(G * order_scores(im, s)); 0/0 -> NaN where a pair has no violation, as autograd."""
    c = np.maximum(np.asarray(s, np.float64)[None, :, :] - np.asarray(im, np.float64)[:, None, :], 0.0)
    nrm = np.sqrt(np.sum(c * c, axis=2))
    with np.errstate(invalid='ignore', divide='ignore'):
        W = np.asarray(G, np.float64) / nrm
        u = W[:, :, None] * c
    return u.sum(axis=1).astype(np.float32), (-u.sum(axis=0)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# loss orchestration (alad/alad_model.py:371-454)
# ----------------------------------------------------------------------------------------------
def forward_loss(img_emb, cap_emb, img_set_sbd, cap_seq_sbd, img_len, cap_len, loss_type,
                 margin=0.2, max_violation=True, aggregation='MrSw'):
    """Ordered dict of loss terms exactly as ALADModel.forward_loss builds it
    (alad/alad_model.py:371-428): sets arrive (S, B, D) and are permuted (:377-378); matching is
    always computed (:380) but only reported when the substring 'matching' is in loss-type (:381);
    alignment is computed when 'alignment' or 'distillation' is requested (:385-390); listnet
    distillation uses the alignment scores as teacher (:404-408)."""
    types = loss_type.split('-')
    im_set = np.transpose(img_set_sbd, (1, 0, 2))
    s_seq = np.transpose(cap_seq_sbd, (1, 0, 2))
    losses = {}
    M = dot_scores(img_emb, cap_emb)
    if 'matching' in loss_type:
        losses['matching'] = hinge_loss(M, margin, max_violation)
    if 'alignment' in types or 'distillation' in types:
        S = alignment_scores(im_set, s_seq, img_len, cap_len, aggregation)
        if 'alignment' in types:
            losses['alignment'] = hinge_loss(S, margin, max_violation)
    if 'distillation' in types:
        losses['distillation'] = listnet_loss(S, M)
    return losses


def total_loss(loss_dict, weights, epoch=0, distill_epoch=2):
    """ALADModel.forward's combination (alad/alad_model.py:442-454), fixed list weights."""
    d = dict(loss_dict)
    if epoch < distill_epoch and len(d) > 1:
        d.pop('distillation', None)
    return sum(d[k] * weights[k] for k in d), d


# ----------------------------------------------------------------------------------------------
# retrieval metrics (alad/recall_auxiliary.py:8-69, alad/evaluation.py:158-327)
# ----------------------------------------------------------------------------------------------
def _metrics(ranks):
    r1 = 100.0 * np.sum(ranks < 1) / len(ranks)
    r5 = 100.0 * np.sum(ranks < 5) / len(ranks)
    r10 = 100.0 * np.sum(ranks < 10) / len(ranks)
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    return r1, r5, r10, medr, meanr


def ranks_from_scores(d_i2t, caps_per_img=5):
    """Ranks for both directions from the (n_img, n_cap) score matrix, defined as the number of
    strictly larger scores (== the argsort position used at recall_auxiliary.py:36-46,52-56 and
    evaluation.py:213-223,303-308 whenever there are no exact ties)."""
    n_img, n_cap = d_i2t.shape
    r_i2t = np.empty(n_img)
    top1_i2t = d_i2t.argmax(1).astype(np.float64)
    for i in range(n_img):
        gt = d_i2t[i, caps_per_img * i:caps_per_img * (i + 1)]
        r_i2t[i] = (d_i2t[i][None, :] > gt[:, None]).sum(1).min()
    gt_col = d_i2t[np.arange(n_cap) // caps_per_img, np.arange(n_cap)]
    r_t2i = (d_i2t > gt_col[None, :]).sum(0).astype(np.float64)
    top1_t2i = d_i2t.argmax(0).astype(np.float64)
    return r_i2t, top1_i2t, r_t2i, top1_t2i


def recall(images, captions, mode='i2t', return_ranks=False, caps_per_img=5):
    """recall(), alad/recall_auxiliary.py:8-69: image rows 0::5 are the distinct images (:14-27);
    i2t ranks the best of the 5 ground-truth captions (:30-46), t2i ranks image k for each of its
    captions (:47-56); R@K, medr, meanr at :61-65."""
    ims = np.asarray(images)[0::caps_per_img]
    d = dot_scores(ims, captions)
    r_i2t, t_i2t, r_t2i, t_t2i = ranks_from_scores(d, caps_per_img)
    ranks, top1 = (r_i2t, t_i2t) if mode == 'i2t' else (r_t2i, t_t2i)
    m = _metrics(ranks)
    return (m, (ranks, top1)) if return_ranks else m


def compute_recall(img_embs, cap_embs):
    """compute_recall -> recall_test, alad/recall_auxiliary.py:72-86,133-149: 7-tuple."""
    r1, r5, r10, _, _ = recall(img_embs, cap_embs, 'i2t')
    r1i, r5i, r10i, _, _ = recall(img_embs, cap_embs, 't2i')
    return r1, r5, r10, r1i, r5i, r10i, r1 + r5 + r10 + r1i + r5i + r10i


def recall_1k_5fold(img_embs, cap_embs):
    """recall_1k_5fold_test, alad/recall_auxiliary.py:90-130: recall_test on five consecutive 5000-row folds
    (torch.split(.., 5000), :99-100), the six recalls averaged over the folds (:117-123)."""
    img_embs, cap_embs = np.asarray(img_embs), np.asarray(cap_embs)
    res = np.array([compute_recall(img_embs[5000 * k:5000 * (k + 1)], cap_embs[5000 * k:5000 * (k + 1)])[:6] for k in range(5)],
                   dtype=np.float64)
    r = [float(v) for v in res.mean(0)]
    return tuple(r) + (sum(r),)


def i2t(images, captions, img_len, cap_len, sim='matching', return_ranks=False,
        aggregation='MrSw'):
    """alad/evaluation.py:158-241.  sim='matching' is the sim_function=None branch (:196, slot-0
    global embeddings); sim='alignment' is the alignment_sim_fn branch (:199-211) whose chunking
    over cap_batches does not change the scores."""
    images = np.asarray(images)
    captions = np.asarray(captions)
    if sim == 'matching':
        d = dot_scores(images[0::5, 0, :], captions[:, 0, :])
    else:
        d = alignment_scores(images[0::5], captions, list(img_len[0::5]), cap_len, aggregation)
    r, t, _, _ = ranks_from_scores(d)
    m = _metrics(r) + (0, 0)
    return (m, (r, t)) if return_ranks else m


def t2i(images, captions, img_len, cap_len, sim='matching', return_ranks=False,
        aggregation='MrSw'):
    """alad/evaluation.py:244-327 (sim_function=None: :285; alignment: :288-301)."""
    images = np.asarray(images)
    captions = np.asarray(captions)
    if sim == 'matching':
        d = dot_scores(images[0::5, 0, :], captions[:, 0, :])
    else:
        d = alignment_scores(images[0::5], captions, list(img_len[0::5]), cap_len, aggregation)
    _, _, r, t = ranks_from_scores(d)
    m = _metrics(r) + (0, 0)
    return (m, (r, t)) if return_ranks else m

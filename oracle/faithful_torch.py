"""CPU baseline: the alignment triplet loss the way the reference EXECUTES it.  TEST INFRASTRUCTURE.

``bench.py``'s ``cpu_baseline`` leg times this on the GPU box's host cores (kind "port": the
reference's own Python files cannot travel to that box).  It restates, op for op, what
``AlignmentContrastiveLoss.forward`` (reference alad/loss.py:79-159, 'MrSw') and
``compute_contrastive_loss`` (alad/loss.py:42-67) ask PyTorch-CPU to do -- broadcast both sets to
(Bi, Bc, ., D), one batched matmul over Bi*Bc tiny problems, boolean length masks, masked_fill,
max over regions, sum over words, VSE++ hinge -- so that its run time is the reference's run time.
``tests/test_oracle_golden.py`` pins it against the golden fixtures made from the reference.
Only tests/, smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F


def alignment_scores_faithful(im_set, s_seq, im_len, s_len, aggregation='MrSw'):
    """(Bi, Bc) scores via the reference's dataflow (alad/loss.py:80-135), differentiable."""
    a = F.normalize(im_set, p=2, dim=2)[:, 1:, :]                  # :80,87
    b = F.normalize(s_seq, p=2, dim=2)[:, 1:-2, :]                 # :81,88
    Bi, Rp = a.shape[0], a.shape[1]
    Bc, Tp = b.shape[0], b.shape[1]
    a4 = a.unsqueeze(1).expand(-1, Bc, -1, -1)                     # :97
    b4 = b.unsqueeze(0).expand(Bi, -1, -1, -1)                     # :98
    al = torch.matmul(a4, b4.permute(0, 1, 3, 2))                  # :99  (Bi, Bc, R', T')
    rvalid = torch.arange(Rp).unsqueeze(0) < (torch.as_tensor(im_len) - 1).unsqueeze(1)
    wvalid = torch.arange(Tp).unsqueeze(0) < (torch.as_tensor(s_len) - 3).unsqueeze(1)
    dead = ~(rvalid[:, None, :, None] & wvalid[None, :, None, :])  # :103-115
    al = al.masked_fill(dead, 0.0)                                 # :116
    if aggregation == 'MrSw':
        return al.max(2)[0].sum(2)                                 # :124-125
    if aggregation == 'MrAVGw':
        return al.max(2)[0].sum(2) / (torch.as_tensor(s_len, dtype=al.dtype) - 3).unsqueeze(0)   # :126-129
    if aggregation == 'MwSr':
        return al.max(3)[0].sum(2)                                 # :134-135
    if aggregation == 'symm':
        return al.max(2)[0].sum(2) + al.max(3)[0].sum(2)           # :130-133
    if aggregation == 'sum':
        return al.sum(dim=(2, 3))                                  # :120-121
    if aggregation == 'mean':
        return al.mean(dim=(2, 3))                                 # :122-123
    if aggregation == 'scan-sentences':                            # :136-149
        nrm = F.normalize(F.relu(al), p=2, dim=2)
        # The reference fills the dead entries with -inf, so a region row r >= Li is all -inf, its
        # softmax NaN, and (the row being zeroed only afterwards, :147) autograd returns NaN for every
        # ragged batch.  Here such rows get finite logits and are zeroed the same way: identical scores,
        # and the gradient of the masked expression instead of NaN.
        dead_row = ~rvalid[:, None, :, None].expand_as(dead)
        logits = nrm.masked_fill(dead & ~dead_row, float('-inf')).masked_fill(dead_row, 0.0)
        w = torch.softmax(logits, dim=3)
        att = torch.matmul(w, b4)                                  # (Bi, Bc, R', D)
        cos = F.cosine_similarity(a4, att, dim=3)
        return cos.masked_fill(~rvalid[:, None, :].expand_as(cos), 0.0).sum(2)
    raise ValueError(aggregation)


def hinge_faithful(scores, margin, max_violation):
    """alad/loss.py:42-67."""
    diag = scores.diag().view(-1, 1)
    cost_s = (margin + scores - diag).clamp(min=0)
    cost_im = (margin + scores - diag.t()).clamp(min=0)
    eye = torch.eye(scores.size(0), dtype=torch.bool)
    cost_s = cost_s.masked_fill(eye, 0)
    cost_im = cost_im.masked_fill(eye, 0)
    if max_violation:
        return cost_s.max(1)[0].sum() + cost_im.max(0)[0].sum()
    return cost_s.sum() + cost_im.sum()


def alignment_triplet_step(im_set, s_seq, im_len, s_len, margin=0.2, max_violation=True,
                           backward=True):
    """One forward (+ backward) of the alignment-head triplet loss; returns (loss, S)."""
    im_set = im_set.detach().requires_grad_(backward)
    s_seq = s_seq.detach().requires_grad_(backward)
    S = alignment_scores_faithful(im_set, s_seq, im_len, s_len)
    loss = hinge_faithful(S, margin, max_violation)
    if backward:
        loss.backward()
        return loss.detach(), S.detach(), im_set.grad, s_seq.grad
    return loss.detach(), S.detach(), None, None

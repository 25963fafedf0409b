"""Drop-in loss modules with the reference's names, constructor arguments and forward signatures
(reference alad/loss.py), computed by the HIP kernels behind include/aladin_hip.h.

    AlignmentContrastiveLoss  <- alad/loss.py:70-159
    ContrastiveLoss           <- alad/loss.py:162-186
    DistillationLoss          <- alad/loss.py:359-447
    Contrastive               <- alad/loss.py:29-67 (shared hinge)

None of them holds parameters or buffers, except DistillationLoss(mode='mse') whose learnable pair
`wb` the reference has too (:366) -- state dicts saved by the reference load unchanged (SURVEY.md
section 5, checkpoint row).  Every aggregation / distillation mode / measure of the reference is
computed by HIP kernels; nothing falls back to eager PyTorch.
"""
import torch
from torch import nn

from . import ops


def l2norm(X):
    """X / sqrt(sum_dim1 X^2), no eps -- reference alad/utils.py:134-139 (zero rows give NaN), HIP kernel."""
    return ops.l2norm_rows(X)


def dot_sim(im, s):
    """reference alad/loss.py:8-11."""
    return ops.dot_scores(im, s)


def cosine_sim(im, s):
    """reference alad/loss.py:13-18."""
    return ops.dot_scores(l2norm(im), l2norm(s))


def order_sim(im, s):
    """reference alad/loss.py:20-26."""
    return ops.order_scores(im, s)


class Contrastive(nn.Module):
    """reference alad/loss.py:29-67."""

    def __init__(self, margin=0, measure=False, max_violation=False):
        super().__init__()
        self.margin = margin
        if measure == 'order':
            self.sim = order_sim
        elif measure == 'cosine':
            self.sim = cosine_sim
        elif measure == 'dot':
            self.sim = dot_sim
        self.max_violation = max_violation

    def compute_contrastive_loss(self, scores):
        return ops.hinge_loss(scores, self.margin, self.max_violation)


class AlignmentContrastiveLoss(Contrastive):
    """reference alad/loss.py:70-159.  aggregation: 'MrSw' (all shipped configs), 'MrAVGw' (= MrSw
    divided by the caption length, :126-129), 'MwSr' / 'symm' (the MrSw kernels with the sets'
    roles swapped, :130-135), 'sum' / 'mean' (:120-123: the double sum of cosines factorises into
    one dot product of the summed unit vectors), 'scan-sentences' (:136-149, fp32 throughout)."""

    def __init__(self, margin=0, measure=False, max_violation=False, aggregation='sum-max-sentences'):
        super().__init__(margin, measure, max_violation)
        self.aggregation = aggregation

    def forward(self, im_set, s_seq, im_len, s_len, return_loss=True, return_similarity_mat=False):
        if self.aggregation not in ('MrSw', 'MrAVGw', 'MwSr', 'symm', 'sum', 'mean', 'scan-sentences'):
            # the reference leaves aggr_similarity unbound and dies with a NameError (:151-159)
            raise ValueError("aladin_amd: unknown alignment aggregation %r (supported: 'MrSw', 'MrAVGw', 'MwSr', "
                             "'symm', 'sum', 'mean', 'scan-sentences')" % (self.aggregation,))
        if return_loss and self.aggregation == 'MrSw':
            # fused scores + hinge node; both outputs are differentiable, as the reference's (ops.alignment_triplet_loss)
            loss, aggr_similarity = ops.alignment_triplet_loss(im_set, s_seq, im_len, s_len, self.margin,
                                                               self.max_violation)
            return (loss, aggr_similarity) if return_similarity_mat else loss
        if self.aggregation == 'scan-sentences':
            aggr_similarity = ops.alignment_scan_scores(im_set, s_seq, im_len, s_len)
        elif self.aggregation in ('sum', 'mean'):
            aggr_similarity = ops.alignment_sum_scores(im_set, s_seq, im_len, s_len, mean=(self.aggregation == 'mean'))
        elif self.aggregation in ('MwSr', 'symm'):
            aggr_similarity = ops.alignment_scores(im_set, s_seq, im_len, s_len, self.aggregation)
        else:
            aggr_similarity = ops.alignment_scores(im_set, s_seq, im_len, s_len)
        if self.aggregation == 'MrAVGw':
            lens = ops.lengths_tensor(s_len, aggr_similarity.device).to(torch.float32) - 3.0
            aggr_similarity = aggr_similarity / lens.unsqueeze(0)
        if return_loss:
            loss = self.compute_contrastive_loss(aggr_similarity)
        if return_loss and return_similarity_mat:
            return loss, aggr_similarity
        elif return_loss:
            return loss
        elif return_similarity_mat:
            return aggr_similarity


class ContrastiveLoss(Contrastive):
    """reference alad/loss.py:162-186."""

    def forward(self, im, s, return_similarity_mat=False):
        if self.sim is dot_sim and getattr(im, 'is_cuda', False) and im.dim() == 2 and tuple(im.shape) == tuple(s.shape):
            # measure 'dot' (every shipped config): scores + hinge behind ONE autograd node (ops.match_hinge)
            loss, scores = ops.match_hinge(im, s, self.margin, self.max_violation)
            return (loss, scores) if return_similarity_mat else loss
        scores = self.sim(im, s)
        loss = self.compute_contrastive_loss(scores)
        if return_similarity_mat:
            return loss, scores
        return loss


class DistillationLoss(nn.Module):
    """reference alad/loss.py:359-447: modes 'mse', 'ordinal', 'contrastive', 'listnet' (the one every
    shipped config uses)."""

    def __init__(self, mode='mse', margin=0.2, threshold=0.1, stride=3):
        super().__init__()
        if mode not in ('mse', 'ordinal', 'contrastive', 'listnet'):
            # the reference would leave `loss` unbound and fail at :447
            raise ValueError('aladin_amd: unknown distillation mode %r' % (mode,))
        self.mode = mode
        self.margin = margin
        self.threshold = threshold
        self.stride = stride
        if mode == 'mse':
            self.wb = nn.Parameter(torch.FloatTensor([0.5, 0.5]), requires_grad=True)     # :366

    def forward(self, teacher_scores, student_scores):
        if self.mode == 'listnet':
            return ops.listnet_loss(teacher_scores, student_scores, temperature=6.0, eps=1e-10)
        return ops.distillation_loss(teacher_scores, student_scores, self.mode, self.margin, self.threshold,
                                     self.stride, wb=self.wb if self.mode == 'mse' else None)

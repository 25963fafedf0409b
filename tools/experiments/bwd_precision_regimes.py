#!/usr/bin/env python3
"""Round 6: how far is the backward's fp16 row step from the reference's autograd OUTSIDE the fixtures?

The row step of the default mode gathers its partner rows -- the unit vectors s^ an output row's gradient is a weighted sum of --
from the forward's packed fp16 operands: an absolute error of <= 2^-11 |s^_c| per component BEFORE the normalise backward
dx = (dx^ - x^ <x^, dx^>) / |x|.  When an image region and the words it pairs with are ALIGNED (cosine near 1: a trained model's
matched pairs) the projection cancels most of dx^ and the rounding error does not shrink with it: relative to the surviving
gradient it grows like 2^-11 / sin(angle).  The reference fixtures and the random bench batch have cosines around 0 ... 0.5; this
probe sweeps the alignment (synth.structured_alignment_batch: noise 3 -> cos 0.1, noise 0.25 -> cos 0.94) at D = 64 and 768 and
prints, per mode, the worst |gradient - oracle| / max |oracle| of the fused training step (oracle: float64 closed form of the
autograd of alad/loss.py:79-159 on the step's own hardest negatives).

    python tools/experiments/bwd_precision_regimes.py            # on the GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)
import numpy as np
import torch

import alad_oracle as O
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss

dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
MODES = ('exact', 'fp16', 'fp16-own')
MARGIN = 100.0      # every row and column violates: the diagonal (matched, ALIGNED) pairs always carry a gradient, as in a trained model's active rows


def step_errors(B, R, Tn, D, noise, ragged, seed):
    im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=seed, noise=noise, ragged=ragged)
    crit = AlignmentContrastiveLoss(MARGIN, 'dot', True, 'MrSw')
    out, ref = {}, None
    for m in MODES:
        ops.set_backward_precision(m)
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss, S = crit(a, b, il, sl, return_similarity_mat=True)
        loss.backward()
        if ref is None:
            _, dS = O.hinge_loss(S.detach().cpu().numpy(), MARGIN, True, return_grad=True)
            ref = O.alignment_scores_backward(im, s, il, sl, dS)
            npairs = int((dS != 0).sum())
        e_max, e_rel = 0.0, 0.0
        for got, want in ((a.grad, ref[0]), (b.grad, ref[1])):
            got = got.cpu().numpy().astype(np.float64)
            scale = max(1e-30, float(np.abs(want).max()))
            err = np.abs(got - want)
            e_max = max(e_max, float(err.max()) / scale)
            big = np.abs(want) >= 0.1 * scale                         # element-relative error on the entries that matter
            if big.any():
                e_rel = max(e_rel, float((err[big] / np.abs(want[big])).max()))
        out[m] = (e_max, e_rel)
    # mean cosine of a matched (region, word) pair, for the record
    x = im[:, 1:] / np.linalg.norm(im[:, 1:], axis=-1, keepdims=True)
    w = s[:, 1:-2] / np.linalg.norm(s[:, 1:-2], axis=-1, keepdims=True)
    cos = float(np.einsum('brd,bwd->brw', x[:4], w[:4]).max(axis=1).mean())
    return out, cos, npairs


if __name__ == '__main__':
    print('worst |grad - oracle| / max|oracle|   [and worst element-relative error over entries >= 0.1 max]   per backward row mode')
    for D in (64, 768):
        for noise in (3.0, 1.0, 0.5, 0.25, 0.1):
            for B, ragged in ((32, True), (128, False)):
                res, cos, npairs = step_errors(B, 34, 50, D, noise, ragged, seed=500 + int(noise * 100) + D)
                print('D=%-4d noise=%-5g B=%-4d ragged=%-5s matched cos~%.2f  pairs=%-4d ' % (D, noise, B, ragged, cos, npairs) +
                      '   '.join('%s %.2e [%.2e]' % (m, res[m][0], res[m][1]) for m in MODES), flush=True)
    ops.set_backward_precision('fp16')

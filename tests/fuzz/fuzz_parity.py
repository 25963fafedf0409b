#!/usr/bin/env python3
"""Randomised parity fuzz (GPU): random shapes, lengths and pooling modes; HIP scores vs the oracle and
HIP gradients vs the oracle chained on the HIP scores.  usage: tests/fuzz/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch

import alad_oracle as O
import faithful_torch as FT
from aladin_amd import ops, synth

# Every backward case draws its row step: 'exact' (raw fp32 rows: 5e-5 of the largest gradient entry on top of rtol 1e-3) or the
# LIBRARY DEFAULT 'fp16' (partner rows from the packed fp16 operands: 5e-4 for D >= 64, 1e-3 for the toy widths below -- the gates of
# tests/test_gpu_parity.py: assert_grads_close).  (the default must be fuzzed here too, not only in fuzz_round3 / 4.)
GRAD_GATE = {'exact': lambda D: 5e-5, 'fp16': lambda D: 5e-4 if D >= 64 else 1e-3}
n_mode = {'exact': 0, 'fp16': 0}
worst_grad = {'exact': 0.0, 'fp16': 0.0}

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
dev = torch.device('cuda:0')
t0 = time.time()
n = nbwd = ties = 0
worst = 0.0
while time.time() - t0 < budget:
    Bi = int(rng.randint(1, 40))
    square = rng.rand() < 0.6
    Bc = Bi if square else int(rng.randint(1, 40))
    R = int(rng.choice([2, 3, 8, 17, 33, 34, 35, 36, 38, 39, 40, 41, 42, 45, 49, 50, 51, 54, 57, 58, 65, 66, 71, 97]))      # 42..57: the 48-row class (round 4)
    Tn = int(rng.choice([4, 5, 9, 11, 12, 19, 20, 25, 27, 28, 35, 36, 37, 39, 41, 43, 44, 50, 51, 67, 68, 71, 99]))      # 11 / 27 / 43: the 8- / 24- / 40-word classes filled
    D = int(rng.choice([8, 24, 64, 100, 128, 768]))
    if Bi * Bc * R * Tn * D > 6e8:
        continue
    case_seed = int(rng.randint(1, 1 << 30))
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, D, seed=case_seed, ragged=True, Bc=Bc)
    il = [int(rng.randint(1, R + 1)) for _ in range(Bi)]
    sl = [int(rng.randint(3, Tn + 1)) for _ in range(Bc)]
    mode = str(rng.choice(['MrSw', 'MrSw', 'MwSr', 'symm', 'sum', 'mean']))
    a = torch.from_numpy(im).to(dev).requires_grad_(True)
    b = torch.from_numpy(s).to(dev).requires_grad_(True)
    if mode in ('sum', 'mean'):
        S = ops.alignment_sum_scores(a, b, il, sl, mean=(mode == 'mean'))
    else:
        S = ops.alignment_scores(a, b, il, sl, mode)
    S_np = S.detach().cpu().numpy()
    ref = O.alignment_scores(im, s, il, sl, mode)
    mag = max(1e-6, float(np.abs(ref).max()))
    err = float(np.abs(S_np - ref).max()) / mag
    worst = max(worst, err)
    tag = 'case Bi=%d Bc=%d R=%d T=%d D=%d mode=%s seed=%d' % (Bi, Bc, R, Tn, D, mode, case_seed)
    # every cosine of two fp16-rounded unit vectors is within 2^-10 of the fp32 value (worst case), so the
    # pooled score is within 2^-10 x (number of pooled terms); at D >= 64 the rounding errors average
    # out and the score must also be within 2e-3 of the score magnitude
    Rq, Tq = R - 1, Tn - 3
    terms = {'MrSw': Tq, 'MwSr': Rq, 'symm': Rq + Tq, 'sum': Rq * Tq, 'mean': 1}[mode]
    abs_err = float(np.abs(S_np - ref).max())
    assert np.isfinite(S_np).all() and abs_err <= 2.0 ** -10 * terms + 1e-6, '%s: score abs error %.3e' % (tag, abs_err)
    # (relative check only where the matrix is large enough for max|S| to be a meaningful scale: a 1x1
    #  score that is a sum of cancelling cosines can be arbitrarily close to zero)
    assert D < 64 or Bi * Bc < 16 or err < 2e-3, '%s: score error %.3e of the score magnitude' % (tag, err)
    n += 1
    if D % 4 == 0 and Bi * Bc * R * Tn < 3e5:
        w = torch.from_numpy(rng.randn(Bi, Bc).astype(np.float32) * (rng.rand(Bi, Bc) < 0.3)).to(dev)
        bwd_mode = str(rng.choice(['exact', 'fp16']))
        ops.set_backward_precision(bwd_mode)
        gate = GRAD_GATE[bwd_mode](D)
        n_mode[bwd_mode] += 1
        tag += ' bwd=' + bwd_mode
        try:
            (S * w).sum().backward()
        except Exception as exc:
            raise AssertionError('%s il=%s sl=%s: backward raised %r' % (tag, il, sl, exc))
        ra, rb = torch.from_numpy(im).requires_grad_(True), torch.from_numpy(s).requires_grad_(True)
        # reference gradients with the argmax structure of the fp32 restatement
        (FT.alignment_scores_faithful(ra, rb, il, sl, mode) * w.cpu()).sum().backward()
        for got, want, nm in ((a.grad, ra.grad, 'd_im'), (b.grad, rb.grad, 'd_s')):
            want = want.numpy()
            got = got.cpu().numpy()
            scale = max(1e-9, float(np.abs(want).max()))
            bad = np.abs(got - want) > 1e-3 * np.abs(want) + gate * scale
            if bad.any():
                # Whole rows move when an argmax flips.  The fp32 torch reference and the kernel's exact
                # fp32 re-decision sum in different orders, so two candidates closer than fp32 rounding may
                # legitimately be ranked differently: accept the case iff the float64 restatement sides
                # with the kernel (anything else is a real error).
                r64a, r64b = torch.from_numpy(im).double().requires_grad_(True), torch.from_numpy(s).double().requires_grad_(True)
                (FT.alignment_scores_faithful(r64a, r64b, il, sl, mode) * w.cpu().double()).sum().backward()
                want64 = (r64a.grad if nm == 'd_im' else r64b.grad).numpy()
                bad64 = np.abs(got - want64) > 1e-3 * np.abs(want64) + gate * scale
                assert not bad64.any(), '%s: %s mismatch in %d elements vs fp32 AND %d vs fp64 reference (max abs %.3e, scale %.3e)' % (
                    tag, nm, int(bad.sum()), int(bad64.sum()), float(np.abs(got - want64).max()), scale)
                ties += 1
            else:
                worst_grad[bwd_mode] = max(worst_grad[bwd_mode], float(np.abs(got - want).max()) / scale)
        nbwd += 1
print('fuzz ok: %d forward cases (%d with backward: %d exact row step, %d fp16 default; %d fp32-reference near-ties resolved by float64), worst score '
      'error %.2e of the score magnitude, worst gradient error / largest entry exact %.2e fp16 %.2e, %.0f s'
      % (n, nbwd, n_mode['exact'], n_mode['fp16'], ties, worst, worst_grad['exact'], worst_grad['fp16'], time.time() - t0))

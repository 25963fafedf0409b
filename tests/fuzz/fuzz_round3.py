#!/usr/bin/env python3
"""Randomised checks (GPU) of what round 3 added:
  bucket    evaluation grids scored in length classes (ops.bucket_plan / GridPlan) vs the single launch and the float64 oracle,
            (N, L, D) tensors and packed stores (bit-equal to each other), random class thresholds;
  partners  the opt-in fp16-partner backward (ops.set_backward_precision('fp16')) vs the oracle's gradients: rtol 1e-3 + 5e-4 of the
            largest entry, and the same zero pattern as the exact path.
  dense     the dense-dS backward (ALADIN_BWD_DENSE: arg-max table from the split-precision tile kernel; row step as two MFMA
            GEMMs): table + gather bit-identical to the per-pair list path, the GEMM row step within 1e-5 of the largest entry
            (5e-4 with fp16 partners), for the sum-of-violations hinge and for arbitrary real dS, random shapes / lengths.
usage: tests/fuzz/fuzz_round3.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch

import alad_oracle as O
from aladin_amd import evaluation as E, ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
from aladin_amd.store import PackedSetStore

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
t0 = time.time()
counts = {'bucket': 0, 'partners': 0, 'dense': 0}
worst = {'bucket_split': 0.0, 'partners': 0.0, 'dense_gemm': 0.0, 'dense_gemm16': 0.0}
real_plan = ops.bucket_plan
ops.DENSE_MIN_FRACTION = 0.0        # the dense path whatever the density of a case's dS
ops.DENSE_GEMM_FORCE = True         # ... and the GEMM row step however full its captions
while time.time() - t0 < budget:
    kind = ['bucket', 'partners', 'dense'][int(rng.randint(0, 3))]
    case_seed = int(rng.randint(1, 1 << 30))
    if kind == 'bucket':
        n_img = int(rng.randint(4, 40))
        D = int(rng.choice([16, 64, 128]))
        L = int(rng.choice([40, 71]))
        lo_i, hi_i = sorted(int(v) for v in rng.randint(3, L + 1, 2))
        lo_c, hi_c = sorted(int(v) for v in rng.randint(4, L + 1, 2))
        images, captions, il, cl = synth.eval_sets(n_img, D, case_seed % 100000, L=L, img_len_range=(lo_i, max(hi_i, lo_i)),
                                                   cap_len_range=(lo_c, max(hi_c, lo_c)), n_full=int(rng.randint(0, 3)))
        ims, ils = images[0::5], il[0::5]
        ops.BUCKET_MIN_PAIRS, ops.BUCKET_MIN_SAMPLES, ops.BUCKET_MIN_GAIN = 1, int(rng.randint(1, 6)), float(rng.choice([0.0, 0.1]))
        for prec in ('split', 'fp16'):
            ops.set_eval_precision(prec)
            ops._PLAN_CACHE.clear()
            ops.bucket_plan = real_plan
            S_b = E.compute_sim_matrix(T(ims), T(captions), ils, cl, mode='alignment')
            si, sc = PackedSetStore(D, 0, dev, capacity_rows=64, precision=prec, padded_len=L), PackedSetStore(D, 2, dev, capacity_rows=64, precision=prec, padded_len=L)
            for k0 in range(0, images.shape[0], 7):
                k1 = min(images.shape[0], k0 + 7)
                si.append(T(images[k0:k1, :max(il[k0:k1])]), il[k0:k1])
                sc.append(T(captions[k0:k1, :max(cl[k0:k1])]), cl[k0:k1])
            S_s = E.compute_sim_matrix(si.view(slice(0, None, 5)), sc, mode='alignment')
            assert torch.equal(S_b, S_s), ('store != dense', case_seed, prec)
            ops.bucket_plan = lambda *a: None
            ops._PLAN_CACHE.clear()
            S_1 = E.compute_sim_matrix(T(ims), T(captions), ils, cl, mode='alignment')
            ref = O.alignment_scores(ims, captions, ils, cl, dtype=np.float64)
            d1 = float((S_b - S_1).abs().max())
            d2 = float(np.abs(S_b.cpu().numpy() - ref).max())
            scale = max(1e-6, float(np.abs(ref).max()))
            if prec == 'split':
                worst['bucket_split'] = max(worst['bucket_split'], d2)
                # a different kernel class sums in a different order: a few fp32 ulps of the score
                assert d1 <= 2e-6 + 1e-6 * scale and d2 <= 4e-6 + 3e-6 * scale, ('bucket split', case_seed, d1, d2, scale)
            else:
                assert d1 <= 1e-5 * scale + 1e-6 and d2 <= 1e-3 * scale, ('bucket fp16', case_seed, d1, d2)
        ops.bucket_plan = real_plan
        ops._PLAN_CACHE.clear()
    elif kind == 'dense':
        B = int(rng.choice([128, 136, 160, 200, 256]))
        R, Tn, D = int(rng.randint(5, 67)), int(rng.randint(6, 51)), int(rng.choice([64, 128, 260, 768]))
        if B * (R + Tn) * D > 2.2e7:
            D = 128
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=case_seed % 100000, noise=float(rng.choice([1.0, 3.0])), ragged=bool(rng.randint(0, 2)))
        il = [max(2, v) for v in il]
        sl = [max(4, v) for v in sl]
        real_dS = bool(rng.randint(0, 2))
        w = torch.randn(B, B, device=dev, generator=torch.Generator(dev).manual_seed(case_seed)) * float(10.0 ** rng.randint(-6, 3))
        grads = {}
        for mode in ('list', 'gather', 'gemm', 'gemm16'):
            ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM = mode != 'list', mode.startswith('gemm')
            ops.set_backward_precision('fp16' if mode == 'gemm16' else 'exact')
            a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
            # the fused triplet node knows when dS is dense: sum of violations, or a gradient on the returned score matrix
            if real_dS:
                loss, S = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
                (loss + (S * w).sum()).backward()
            else:
                AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')(a, b, il, sl).backward()
            grads[mode] = (a.grad.clone(), b.grad.clone())
        ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM = True, True
        ops.set_backward_precision('exact')
        for k in (0, 1):
            ref = grads['list'][k]
            scale = max(1e-30, float(ref.abs().max()))
            assert torch.equal(ref, grads['gather'][k]), ('dense table != list', case_seed, B, R, Tn, D, real_dS)
            for mode, tol in (('gemm', 1e-5), ('gemm16', 5e-4)):
                err = float((grads[mode][k] - ref).abs().max()) / scale
                worst['dense_' + mode] = max(worst['dense_' + mode], err)
                # fp16 partners: the opt-in's tolerance (rtol 1e-3 + 5e-4 of the largest entry, as in `partners`)
                assert bool(((grads[mode][k] - ref).abs() <= (1e-3 * ref.abs() if mode == 'gemm16' else 0) + tol * scale).all()), ('dense ' + mode, case_seed, B, R, Tn, D, real_dS, err)
                assert torch.isfinite(grads[mode][k]).all()
    else:
        B = int(rng.choice([3, 8, 17, 40, 64, 96]))
        R, Tn, D = int(rng.randint(5, 67)), int(rng.randint(6, 51)), int(rng.choice([64, 128, 768]))
        if B * (R + Tn) * D > 9e6:
            continue
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=case_seed % 100000, noise=float(rng.choice([1.0, 3.0])), ragged=bool(rng.randint(0, 2)))
        il = [max(2, v) for v in il]
        sl = [max(4, v) for v in sl]
        mv = bool(rng.randint(0, 2))
        grads = {}
        for mode in ('fp16', 'exact'):
            ops.set_backward_precision(mode)
            a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
            S = ops.alignment_scores(a, b, il, sl)
            ops.hinge_loss(S, 0.2, mv).backward()
            grads[mode] = (a.grad.cpu().numpy(), b.grad.cpu().numpy(), S.detach().cpu().numpy())
        ops.set_backward_precision('exact')
        _, dS = O.hinge_loss(grads['fp16'][2], 0.2, mv, return_grad=True)
        dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
        # The opt-in changes the partner VALUES only, the arg-maxima are the exact path's: judged against the exact kernel path
        # (an fp32 near-tie that the float64 oracle resolves the other way swaps a whole partner row in BOTH modes; the exact
        # path's own agreement with the oracle is fuzz_parity.py's subject, which sets such ties aside).
        for got, ex, ref in ((grads['fp16'][0], grads['exact'][0], dim), (grads['fp16'][1], grads['exact'][1], ds)):
            scale = max(1e-9, float(np.abs(ex).max()))
            err = float(np.abs(got - ex).max()) / scale
            worst['partners'] = max(worst['partners'], err)
            assert np.all(np.abs(got - ex) <= 1e-3 * np.abs(ex) + 5e-4 * scale), ('partners', case_seed, B, R, Tn, D, mv, err)
            assert np.array_equal(got == 0, ex == 0) or np.abs(got[(got == 0) != (ex == 0)]).max() <= 1e-6 * scale, ('zero pattern', case_seed)
            if float(np.abs(ex - ref).max()) <= 1e-4 * scale:          # no near-tie swap in this case: the oracle applies too
                assert np.all(np.abs(got - ref) <= 1e-3 * np.abs(ref) + 5e-4 * scale), ('partners vs oracle', case_seed, B, R, Tn, D, mv)
    counts[kind] += 1
print('fuzz_round3 ok:', counts, 'worst', {k: float('%.3g' % v) for k, v in worst.items()}, 'seed', seed)

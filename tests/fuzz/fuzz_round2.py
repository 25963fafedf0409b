#!/usr/bin/env python3
"""Randomised checks (GPU) of what round 2 added:
  split     alignment scores in the rank-exact evaluation precision vs the float64 oracle, every pooling mode with a max,
            trimmed / padded sets, tensors and packed stores (bit-equal to each other);
  heads     the three-launch small-batch loss heads (ops.small_batch_loss_heads) vs the separate differentiable pieces;
  topk      aladin_topk vs a numpy stable argsort, both orientations;
  sgemm     the split-K fp32 MFMA GEMM vs float64;
  fused     the fused hinge + argmax training path (pairs from the hinge statistics, aladin_hinge_argmax_fused /
            aladin_align_bwd_rows) vs the list-driven path and vs the oracle's gradients, any B, margin and raggedness.
usage: tests/fuzz/fuzz_round2.py [seconds] [seed]"""
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
from aladin_amd.store import PackedSetStore

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
t0 = time.time()
counts = {'split': 0, 'heads': 0, 'topk': 0, 'sgemm': 0, 'fused': 0}
worst_split = 0.0
while time.time() - t0 < budget:
    kind = ['split', 'split', 'heads', 'topk', 'sgemm', 'fused'][int(rng.randint(0, 6))]
    case_seed = int(rng.randint(1, 1 << 30))
    if kind == 'split':
        Bi, Bc = int(rng.randint(1, 30)), int(rng.randint(1, 30))
        L = int(rng.choice([5, 12, 34, 40, 50, 71]))
        D = int(rng.choice([8, 24, 64, 100, 768]))
        if Bi * Bc * L * L * D > 4e8:
            continue
        im = synth.normal((Bi, L, D), case_seed)
        s = synth.normal((Bc, L, D), case_seed + 1)
        cap_r = int(rng.randint(2, L + 1))                       # longest real lengths, possibly below the padded length
        cap_t = int(rng.randint(4, L + 1))
        il = [int(rng.randint(1, cap_r + 1)) for _ in range(Bi)]
        sl = [int(rng.randint(3, cap_t + 1)) for _ in range(Bc)]
        for i, n_ in enumerate(il):
            im[i, n_:] = 0
        for j, n_ in enumerate(sl):
            s[j, n_:] = 0
        mode = str(rng.choice(['MrSw', 'MrSw', 'MwSr', 'symm']))
        ref = O.alignment_scores(im, s, il, sl, mode, dtype=np.float64)
        S = ops.alignment_scores(T(im), T(s), il, sl, mode, precision='split').cpu().numpy()
        err = float(np.abs(S - ref).max())
        tag = 'split Bi=%d Bc=%d L=%d D=%d mode=%s seed=%d' % (Bi, Bc, L, D, mode, case_seed)
        assert np.isfinite(S).all() and err <= 4e-6 + 2e-6 * float(np.abs(ref).max()), '%s: abs error %.3e' % (tag, err)
        worst_split = max(worst_split, err)
        if mode == 'MrSw' and L >= 4:
            # packed split stores give the same bits as the tensors
            si = PackedSetStore(D, 0, dev, capacity_rows=8, precision='split', padded_len=L)
            sc = PackedSetStore(D, 2, dev, capacity_rows=8, precision='split', padded_len=L)
            si.append(T(im), il)
            sc.append(T(s), sl)
            S2 = E.compute_sim_matrix(si, sc, mode='alignment').cpu().numpy()
            assert np.array_equal(S2, S), '%s: store and tensor scores differ (max %.3e)' % (tag, float(np.abs(S2 - S).max()))
    elif kind == 'heads':
        B = int(rng.randint(1, 65))
        R, Tn = int(rng.choice([5, 20, 34, 40, 43, 51, 57])), int(rng.choice([6, 11, 24, 27, 38, 43, 50]))
        D = int(rng.choice([64, 128, 768]))
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=case_seed, noise=float(rng.choice([1.0, 3.0])), ragged=True)
        ge, gc = synth.global_embeddings(B, D, seed=case_seed + 5, noise=float(rng.choice([0.5, 1.5])))
        heads = [h for h in ('matching', 'alignment', 'distillation') if rng.rand() < 0.7] or ['alignment']
        mv = bool(rng.rand() < 0.7)
        weights = {'matching': float(rng.rand()), 'alignment': 1.0, 'distillation': float(0.5 + rng.rand())}
        t1 = [T(ge).requires_grad_(True), T(gc).requires_grad_(True), T(im).requires_grad_(True), T(s).requires_grad_(True)]
        total, terms, _, _ = ops.small_batch_loss_heads(t1[0], t1[1], t1[2], t1[3], il, sl, 0.2, mv, heads, weights)
        total.backward()
        t2 = [T(ge).requires_grad_(True), T(gc).requires_grad_(True), T(im).requires_grad_(True), T(s).requires_grad_(True)]
        ref = 0
        if 'alignment' in heads or 'distillation' in heads:
            la, S0 = ops.alignment_triplet_loss(t2[2], t2[3], il, sl, 0.2, mv)
        if 'matching' in heads or 'distillation' in heads:
            lm, ld, _ = ops.small_batch_match_distill(t2[0], t2[1], S0 if 'distillation' in heads else None, 0.2, mv,
                                                      want_hinge='matching' in heads)
        if 'matching' in heads:
            ref = ref + lm * weights['matching']
        if 'alignment' in heads:
            ref = ref + la * weights['alignment']
        if 'distillation' in heads:
            ref = ref + ld * weights['distillation']
        ref.backward()
        tag = 'heads B=%d R=%d T=%d D=%d %s mv=%s seed=%d' % (B, R, Tn, D, heads, mv, case_seed)
        assert abs(float(total) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), '%s: total %r vs %r' % (tag, float(total), float(ref))
        for a, b in zip(t1, t2):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0, tag
            else:
                sc_ = max(1e-12, float(b.grad.abs().max()))
                assert float((a.grad - b.grad).abs().max()) <= 2e-6 * sc_ + 1e-5 * float(b.grad.abs().mean()), tag
    elif kind == 'fused':
        B = int(rng.choice([1, 2, 3, 5, 8, 17, 31, 32, 33, 64, 65, 70, 100]))
        R, Tn = int(rng.choice([3, 9, 20, 33, 34, 42, 49, 51, 57])), int(rng.choice([5, 10, 12, 22, 36, 38, 50, 66]))
        D = int(rng.choice([64, 128, 768]))
        margin = float(rng.choice([0.0, 0.05, 0.2, 1.0, 50.0]))
        # (noise >= 1: with less the regions of an image are nearly parallel and which of them is the fp32 argmax is a matter
        #  of summation order -- the reference's own autograd is ambiguous there)
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=case_seed, noise=float(rng.choice([1.0, 3.0])), ragged=True)
        if rng.rand() < 0.3 and B > 2:
            im[1] = im[0]; s[1] = s[0]; il[1] = il[0]; sl[1] = sl[0]      # duplicated samples: exact ties in S
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss, S = ops.alignment_triplet_loss(a, b, il, sl, margin, True)
        loss.backward()
        ilt, slt = ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev)
        with torch.no_grad():
            S2, packed = ops._align_forward(a.detach(), b.detach(), ilt, slt)
            loss2, dS2, pairs = ops._hinge_raw(S2, margin, True, True, want_pairs=True)
            d_im, d_s = ops._align_backward(a.detach(), b.detach(), ilt, slt, dS2, gscale=torch.ones((), device=dev), packed=packed, pairs=pairs)
        tag = 'fused B=%d R=%d T=%d D=%d margin=%g seed=%d' % (B, R, Tn, D, margin, case_seed)
        assert torch.equal(S, S2) and float(loss) == float(loss2), tag
        assert torch.equal(a.grad, d_im) and torch.equal(b.grad, d_s), tag
        S_np = S.detach().cpu().numpy()
        _, dS_o = O.hinge_loss(S_np, margin, True, return_grad=True)
        assert np.array_equal(dS_o, dS2.cpu().numpy()), tag
        gi, gs = O.alignment_scores_backward(im, s, il, sl, dS_o)
        for got, want in ((a.grad.cpu().numpy(), gi), (b.grad.cpu().numpy(), gs)):
            sc_ = max(1e-3, float(np.abs(want).max()))     # (duplicated samples cancel exactly: only rounding noise is left)
            if float(np.abs(got - want).max()) > 1e-3 * sc_:
                # a whole row moves when an argmax flips; legitimate only if some differentiated (pair, word) has its two
                # best regions closer than fp32 can resolve (the kernel's exact fp32 dot and numpy sum in different orders)
                gap = np.inf
                for i_, j_ in zip(*np.nonzero(dS_o)):
                    Li_, Lj_ = il[i_] - 1, sl[j_] - 3
                    if Li_ < 2 or Lj_ < 1:
                        continue
                    x64, y64 = im[i_, 1:1 + Li_].astype(np.float64), s[j_, 1:1 + Lj_].astype(np.float64)
                    A64 = (x64 / np.linalg.norm(x64, axis=1, keepdims=True)) @ (y64 / np.linalg.norm(y64, axis=1, keepdims=True)).T
                    top = np.sort(A64, axis=0)[-2:]
                    gap = min(gap, float((top[1] - top[0]).min()))
                    if Li_ < R - 1:
                        gap = min(gap, float(np.abs(top[1]).min()))            # the zero fill competes too
                assert gap < 5e-7, '%s: gradient mismatch %.3e with no fp32-unresolvable argmax (smallest top-2 gap %.3e)' % (
                    tag, float(np.abs(got - want).max()), gap)
                counts['near_ties'] = counts.get('near_ties', 0) + 1
    elif kind == 'topk':
        n_q, n_c, k = int(rng.randint(1, 60)), int(rng.randint(1, 3000)), int(rng.choice([1, 5, 50, 64]))
        M = rng.randn(n_q, n_c).astype(np.float32)
        if rng.rand() < 0.3:
            M = np.round(M * 4) / 4                              # many exact ties: lower index first
        dim = int(rng.randint(0, 2))
        X = T(M if dim == 1 else M.T.copy())
        got = ops.topk_indices(X, k, dim=dim).cpu().numpy()
        order = np.argsort(-M, axis=1, kind='stable')[:, :k]
        want = np.full((n_q, k), -1, dtype=np.int64)
        want[:, :min(k, n_c)] = order[:, :min(k, n_c)]
        assert np.array_equal(got, want), 'topk n_q=%d n_c=%d k=%d dim=%d seed=%d' % (n_q, n_c, k, dim, case_seed)
    else:
        Mm, Nn, Kk = int(rng.randint(1, 200)), int(rng.randint(1, 200)), int(rng.choice([1, 7, 32, 33, 100, 768, 1000]))
        A = rng.randn(Mm, Kk).astype(np.float32)
        Bm = rng.randn(Nn, Kk).astype(np.float32)
        got = ops.dot_scores(T(A), T(Bm)).cpu().numpy()
        want = A.astype(np.float64) @ Bm.astype(np.float64).T
        assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.sqrt(Kk))), 'sgemm %dx%dx%d seed=%d' % (Mm, Nn, Kk, case_seed)
    counts[kind] += 1
print('fuzz ok: %s cases, worst split-precision score error %.2e (absolute), %.0f s' % (counts, worst_split, time.time() - t0))

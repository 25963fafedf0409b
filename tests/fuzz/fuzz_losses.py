#!/usr/bin/env python3
"""Randomised parity fuzz (GPU) for the B x B loss kernels and the retrieval kernels against the
oracle.  usage: tests/fuzz/fuzz_losses.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch

import alad_oracle as O
from aladin_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    B = int(rng.choice([1, 2, 3, 5, 16, 31, 64, 100, 255, 256, 257, 600]))
    scale = float(rng.choice([0.05, 1.0, 8.0]))
    S = (rng.randn(B, B) * scale + rng.randn() + np.eye(B) * rng.rand() * 2 * scale).astype(np.float32)
    margin = float(rng.choice([0.0, 0.2, 0.3, 1.0]))
    for mv in (True, False):
        St = T(S).requires_grad_(True)
        loss = ops.hinge_loss(St, margin, mv)
        loss.backward()
        ref, dS = O.hinge_loss(S.astype(np.float64), margin, mv, return_grad=True)
        assert abs(loss.item() - ref) <= 2e-5 * max(1.0, abs(ref)), ('hinge', B, mv, loss.item(), ref)
        # exact ties between costs can pick a different hardest negative: compare only when margins are clear
        got = St.grad.cpu().numpy()
        if not np.array_equal(got, dS):
            _, dS32 = O.hinge_loss(S, margin, mv, return_grad=True)
            assert np.array_equal(got, dS32), ('hinge grad', B, mv, int((got != dS32).sum()))
    if B <= 300:
        teacher = (rng.randn(B, B) * 2 + 3).astype(np.float32)
        student = np.clip(rng.randn(B, B) * 0.3, -1, 1).astype(np.float32)
        st = T(student).requires_grad_(True)
        loss = ops.listnet_loss(T(teacher), st)
        loss.backward()
        ref, dM = O.listnet_loss(teacher, student, return_grad=True)
        assert abs(loss.item() - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), ('listnet', B, loss.item(), float(ref))
        np.testing.assert_allclose(st.grad.cpu().numpy(), dM, rtol=2e-3, atol=2e-6)
    if B >= 2:
        # the other distillation modes: teacher offsets move the ordinal threshold through the data
        teacher = (rng.randn(B, B) * float(rng.choice([0.3, 2.0])) + float(rng.choice([-3.0, 0.0, 3.0]))).astype(np.float32)
        student = (rng.randn(B, B) * float(rng.choice([0.2, 1.5]))).astype(np.float32)
        dm = float(rng.choice([0.0, 0.2, 0.7]))
        thr = float(rng.choice([-1.0, 0.1, 2.5]))
        stride = int(rng.randint(1, min(B, 6)))
        wbv = rng.randn(2).astype(np.float32)
        wb = T(wbv).requires_grad_(True)
        for mode, ref_fn in (('mse', lambda: O.distill_mse(teacher, student, wbv, True)),
                             ('contrastive', lambda: O.distill_contrastive(teacher, student, dm, True)),
                             ('ordinal', lambda: O.distill_ordinal(teacher, student, dm, thr, stride, True))):
            st = T(student).requires_grad_(True)
            loss = ops.distillation_loss(T(teacher), st, mode, dm, thr, stride, wb=wb if mode == 'mse' else None)
            loss.backward()
            out = ref_fn()
            if np.isnan(out[0]):
                assert np.isnan(loss.item()), (mode, B)
            else:
                assert abs(loss.item() - float(out[0])) <= 3e-5 * max(1.0, abs(float(out[0]))), (mode, B, loss.item(), float(out[0]))
            got = st.grad.cpu().numpy()
            if mode == 'mse':
                np.testing.assert_allclose(got, out[1], rtol=1e-4, atol=1e-7)
                np.testing.assert_allclose(wb.grad.cpu().numpy(), out[2], rtol=2e-4, atol=1e-5)
                wb.grad = None
            else:
                # integer-valued (contrastive) / count-normalised (ordinal) gradients; a hinge sitting on
                # the fp32 rounding edge may flip, which moves single entries only
                bad = np.abs(got - out[1]) > 1e-5 * max(1.0, np.abs(out[1]).max())
                assert bad.sum() <= max(2, B // 64), (mode, B, int(bad.sum()))
        Bi2, Bc2, D2 = int(rng.randint(1, 90)), int(rng.randint(1, 90)), int(rng.choice([1, 5, 64, 100, 768]))
        im_o = rng.randn(Bi2, D2).astype(np.float32)
        s_o = (rng.randn(Bc2, D2) + 0.3).astype(np.float32)
        a_o, b_o = T(im_o).requires_grad_(True), T(s_o).requires_grad_(True)
        sc = ops.order_scores(a_o, b_o)
        G = rng.randn(Bi2, Bc2).astype(np.float32)
        (sc * T(G)).sum().backward()
        np.testing.assert_allclose(sc.detach().cpu().numpy(), O.order_scores(im_o, s_o), rtol=5e-6, atol=1e-6)
        di, ds = O.order_scores_backward(im_o, s_o, G)
        if np.isfinite(di).all():
            tol = 3e-5 * max(1.0, float(np.abs(di).max()), float(np.abs(ds).max()))
            np.testing.assert_allclose(a_o.grad.cpu().numpy(), di, rtol=1e-4, atol=tol)
            np.testing.assert_allclose(b_o.grad.cpu().numpy(), ds, rtol=1e-4, atol=tol)
    M, N, K = int(rng.randint(1, 200)), int(rng.randint(1, 200)), int(rng.choice([1, 7, 64, 100, 768]))
    A = rng.randn(M, K).astype(np.float32)
    Bm = rng.randn(N, K).astype(np.float32)
    a, b = T(A).requires_grad_(True), T(Bm).requires_grad_(True)
    C = ops.dot_scores(a, b)
    W = rng.randn(M, N).astype(np.float32)
    (C * T(W)).sum().backward()
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    np.testing.assert_allclose(C.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.sqrt(K))
    np.testing.assert_allclose(a.grad.cpu().numpy(), W.astype(np.float64) @ Bm, rtol=1e-4, atol=1e-4 * np.sqrt(N))
    np.testing.assert_allclose(b.grad.cpu().numpy(), W.astype(np.float64).T @ A, rtol=1e-4, atol=1e-4 * np.sqrt(M))
    n_img = int(rng.choice([1, 2, 7, 50, 255, 700]))
    D = int(rng.choice([8, 64, 100, 768]))
    img = rng.randn(n_img, D).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    cap = np.repeat(img, 5, axis=0) + float(rng.choice([0.5, 3.0, 10.0])) * rng.randn(5 * n_img, D).astype(np.float32) / np.sqrt(D)
    cap = (cap / np.linalg.norm(cap, axis=1, keepdims=True)).astype(np.float32)
    sim = ops.sim_matrix(T(img), T(cap))
    ref = img.astype(np.float64) @ cap.astype(np.float64).T
    assert float(np.abs(sim.cpu().numpy() - ref).max()) < 5e-6, ('sim', n_img, D)
    r_i2t, t_i2t, r_t2i, t_t2i = (x.cpu().numpy() for x in ops.recall_ranks(sim))
    sim_h = sim.cpu().numpy()                              # ranks are checked on the device's own scores: exact
    e_i2t, e_top_i, e_t2i, e_top_t = O.ranks_from_scores(sim_h)
    assert np.array_equal(r_i2t, e_i2t) and np.array_equal(r_t2i, e_t2i), ('ranks', n_img, D)
    assert np.array_equal(sim_h[np.arange(n_img), t_i2t], sim_h.max(1)) and np.array_equal(sim_h[t_t2i, np.arange(5 * n_img)], sim_h.max(0))
    fused = [x.cpu().numpy() for x in ops.retrieval_ranks(T(img), T(cap))]      # no score matrix: same ints
    for got, want, what in zip(fused, (r_i2t, t_i2t, r_t2i, t_t2i), ('r_i2t', 'top_i2t', 'r_t2i', 'top_t2i')):
        assert np.array_equal(got, want), ('fused ' + what, n_img, D)
    n += 1
print('fuzz_losses ok: %d rounds, %.0f s' % (n, time.time() - t0))

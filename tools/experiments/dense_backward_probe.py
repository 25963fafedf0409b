#!/usr/bin/env python3
"""Probes of the dense-dS backward (max_violation=False), one script, two subcommands:

    fill   gather vs GEMM row step as a function of how full the captions are (B = 256, R x T = 34 x 50 and 51 x 38):
           the crossover ops._gemm_rows_pay steers by
    err    gradients of the GEMM row step / the fp16-partner forms against the exact fp32 gather (max error relative to the
           largest entry, zero pattern)

usage: tools/dense_backward_probe.py fill | err [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss


def fill():
    ops.DENSE_MIN_FRACTION = 0.0
    B, D = 256, 768
    for R, T_ in ((34, 50), (51, 38)):
        im, s, il, sl = synth.alignment_batch(B, R, T_, D, seed=3, ragged=False)
        crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
        Tq = T_ - 3
        for frac in (1.0, 0.8, 0.6, 0.45, 0.3):
            L = max(1, int(round(frac * Tq)))
            sl2 = [L + 3] * B
            out = []
            for mode in ('gather', 'gemm'):
                ops.DENSE_ROWS_GEMM = mode == 'gemm'
                ops.DENSE_GEMM_FORCE = True
                a = torch.from_numpy(im).cuda().requires_grad_(True)
                b = torch.from_numpy(s).cuda().requires_grad_(True)

                def step():
                    a.grad = None
                    b.grad = None
                    crit(a, b, il, sl2).backward()
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                out.append((time.perf_counter() - t0) / 5 * 1e3)
            print('R %d T %d  words %d/%d (fill of the %d-word tile %.2f)  gather %.3f ms  gemm %.3f ms'
                  % (R, T_, L, Tq, (Tq + 15) // 16 * 16, L / ((Tq + 15) // 16 * 16), out[0], out[1]), flush=True)


def err(B):
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=B + 5, ragged=True)
    crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
    ops.DENSE_MIN_FRACTION = 0.0
    ops.DENSE_GEMM_FORCE = True
    g = {}
    for mode in ('gather', 'gemm', 'gemm16', 'gather16'):
        a = torch.from_numpy(im).cuda().requires_grad_(True)
        b = torch.from_numpy(s).cuda().requires_grad_(True)
        ops.DENSE_ROWS_GEMM = mode.startswith('gemm')
        ops.set_backward_precision('fp16' if mode.endswith('16') else 'exact')
        crit(a, b, il, sl).backward()
        g[mode] = (a.grad.cpu().numpy().astype(np.float64), b.grad.cpu().numpy().astype(np.float64))
    for mode in ('gemm', 'gemm16', 'gather16'):
        for k, nm in ((0, 'd_im'), (1, 'd_s')):
            ref, got = g['gather'][k], g[mode][k]
            print(mode, nm, 'max|ref| %.3e  max err %.3e  rel-to-max %.3e  zero-pattern equal %s nan %d' % (
                np.abs(ref).max(), np.abs(ref - got).max(), np.abs(ref - got).max() / np.abs(ref).max(),
                np.array_equal(ref == 0, got == 0), int(np.isnan(got).sum())))


if __name__ == '__main__':
    cmd = sys.argv[1] if len(sys.argv) > 1 else 'fill'
    if cmd == 'fill':
        fill()
    elif cmd == 'err':
        err(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
    else:
        raise SystemExit(__doc__)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=B + 5, ragged=True)
crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
ops.DENSE_MIN_FRACTION = 0.0
ops.DENSE_GEMM_FORCE = True
g = {}
for mode in ('gather', 'gemm', 'gemm16', 'gather16'):
    a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
    ops.DENSE_ROWS_GEMM = mode.startswith('gemm')
    ops.set_backward_precision('fp16' if mode.endswith('16') else 'exact')
    crit(a, b, il, sl).backward()
    g[mode] = (a.grad.cpu().numpy().astype(np.float64), b.grad.cpu().numpy().astype(np.float64))
for mode in ('gemm', 'gemm16', 'gather16'):
    for k, nm in ((0, 'd_im'), (1, 'd_s')):
        ref, got = g['gather'][k], g[mode][k]
        print(mode, nm, 'max|ref| %.3e  max err %.3e  rel-to-max %.3e  zero-pattern equal %s nan %d' % (
            np.abs(ref).max(), np.abs(ref - got).max(), np.abs(ref - got).max() / np.abs(ref).max(), np.array_equal(ref == 0, got == 0), np.isnan(got).sum()))

"""gather vs GEMM row step of the dense backward as a function of how full the captions are (B = 256)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
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
            if hasattr(ops, 'DENSE_GEMM_FORCE'): ops.DENSE_GEMM_FORCE = True
            a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
            for _ in range(2): a.grad = None; b.grad = None; crit(a, b, il, sl2).backward()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): a.grad = None; b.grad = None; crit(a, b, il, sl2).backward()
            torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 5 * 1e3)
        print('R %d T %d  words %d/%d (fill of the %d-word tile %.2f)  gather %.3f ms  gemm %.3f ms' % (R, T_, L, Tq, (Tq + 15) // 16 * 16, L / ((Tq + 15) // 16 * 16), out[0], out[1]), flush=True)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
ops.DENSE_MIN_FRACTION = 0.0
ops.DENSE_GEMM_FORCE = True
for B, R, T_, D in ((256, 51, 38, 768), (512, 34, 50, 768), (384, 20, 30, 512), (1024, 34, 50, 256)):
    im, s, il, sl = synth.alignment_batch(B, R, T_, D, seed=B, ragged=True)
    crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
    g = {}
    for mode in ('list', 'gather', 'gemm'):
        ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM = mode != 'list', mode == 'gemm'
        a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
        crit(a, b, il, sl).backward(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            a.grad = None; b.grad = None
            crit(a, b, il, sl).backward()
        torch.cuda.synchronize()
        g[mode] = (a.grad, b.grad, (time.perf_counter() - t0) / 3 * 1e3)
    print(B, R, T_, D, 'ms list %.2f gather %.2f gemm %.2f' % (g['list'][2], g['gather'][2], g['gemm'][2]),
          'table==list', torch.equal(g['list'][0], g['gather'][0]) and torch.equal(g['list'][1], g['gather'][1]),
          'gemm err %.2e %.2e' % (float((g['gemm'][0] - g['list'][0]).abs().max() / g['list'][0].abs().max()),
                                  float((g['gemm'][1] - g['list'][1]).abs().max() / g['list'][1].abs().max())), flush=True)

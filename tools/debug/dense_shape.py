import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
ops.DENSE_MIN_FRACTION = 0.0
ops.DENSE_GEMM_FORCE = True
B, R, T_, D = [int(v) for v in sys.argv[1:5]]
im, s, il, sl = synth.alignment_batch(B, R, T_, D, seed=B, ragged=True)
crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
g = {}
for mode in ('list', 'gather', 'gemm'):
    ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM = mode != 'list', mode == 'gemm'
    a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
    crit(a, b, il, sl).backward(); torch.cuda.synchronize()
    g[mode] = (a.grad, b.grad)
    print(mode, 'flags', ops._LAST_BWD_FLAGS[0])
print('gather==list', torch.equal(g['list'][0], g['gather'][0]), torch.equal(g['list'][1], g['gather'][1]),
      'gemm==gather', torch.equal(g['gemm'][0], g['gather'][0]),
      'gemm err', float((g['gemm'][0] - g['list'][0]).abs().max() / g['list'][0].abs().max()), float((g['gemm'][1] - g['list'][1]).abs().max() / g['list'][1].abs().max()))

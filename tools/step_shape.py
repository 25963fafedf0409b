"""the hardest-negative triplet step (fwd + bwd) at an arbitrary set shape, with the kernel breakdown under rocprofv3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aladin_amd import synth
from aladin_amd.loss import AlignmentContrastiveLoss
B, R, T_, D = [int(v) for v in sys.argv[1:5]]
ragged = len(sys.argv) > 5 and sys.argv[5] == 'ragged'
im, s, il, sl = synth.alignment_batch(B, R, T_, D, seed=B, ragged=ragged)
crit = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')
a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
def step():
    a.grad = None; b.grad = None
    crit(a, b, il, sl).backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print('B %d R %d T %d D %d ragged %s: %.4f ms per step' % (B, R, T_, D, ragged, (time.perf_counter() - t0) / 50 * 1e3))

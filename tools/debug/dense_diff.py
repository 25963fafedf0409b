import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=B + 5, ragged=True)
crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
g = {}
for dense in (False, True):
    a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
    ops.DENSE_BACKWARD = dense
    crit(a, b, il, sl).backward()
    g[dense] = (a.grad.cpu().numpy(), b.grad.cpu().numpy())
da = np.abs(g[True][0] - g[False][0]).max(-1); db = np.abs(g[True][1] - g[False][1]).max(-1)
print('images x regions differing', np.argwhere(da > 0)[:20].tolist(), (da > 0).sum())
print('captions x words differing', np.argwhere(db > 0)[:20].tolist(), (db > 0).sum())
imn = im / np.linalg.norm(im, axis=-1, keepdims=True); sn = s / np.linalg.norm(s, axis=-1, keepdims=True)
print('il', il[:8], 'sl', sl[:8])
for (c, w) in np.argwhere(db > 0)[:6]:
    # which images: those whose region grads differ
    for i in sorted(set(np.argwhere(da > 0)[:, 0].tolist()))[:40]:
        Li = il[i] - 1
        sc = imn[i, 1:].astype(np.float64) @ sn[c, w].astype(np.float64)
        sc[Li:] = 0
        o = np.argsort(-sc)
        gap = sc[o[0]] - sc[o[1]]
        if gap < 1e-4: print('cap', c, 'word(raw idx)', w, 'img', i, 'Li', Li, 'top', o[:2], 'gap', gap, sc[o[0]])

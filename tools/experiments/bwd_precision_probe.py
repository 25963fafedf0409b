#!/usr/bin/env python3
"""Round 5, VERDICT r4 item 2 (i) + (iii): which unit vectors may the backward's row kernel take from the forward's packed fp16
operands?  For every reference fixture with gradients (tests/golden/align_*.npz, both dS kinds) the worst |gradient - reference|
over the largest reference entry, for
    exact      raw fp32 rows everywhere (rounds 1-4)
    partners   partner rows from the packed operands (ALADIN_BWD_PARTNERS_FP16), the output row's own from the raw set
    both       + the output row's own unit vector and inverse norm from the packed operands (ALADIN_BWD_OWN_ROW_FP16)
and the row kernel's time at B = 256 (the fused step's backward call, event-timed).  The gate: <= 5e-4 (half of north_star's 1e-3)
on EVERY fixture."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

from aladin_amd import ops, synth
from conftest import SQUARE_ALIGN_GOLDENS, golden_alignment_inputs, load_golden

dev = torch.device('cuda:0')
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
MODES = {'exact': 'exact', 'partners': 'fp16', 'both': 'fp16-own'}


def setmode(m):
    ops.set_backward_precision(MODES[m])


worst = {m: (0.0, '') for m in MODES}
for name in SQUARE_ALIGN_GOLDENS:
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    a, b, ilt, slt = T(im), T(s), ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev)
    geom = ops.align_geometry(a.shape[0], b.shape[0], a.shape[1], b.shape[1], a.shape[2])
    packed = ops.pack_sets(a, b, ilt, slt, geom)
    st = int(g['grad_stride'])
    for tag in ('mv', 'sum'):
        row = []
        for m in MODES:
            setmode(m)
            d_im, d_s = ops._align_backward(a, b, ilt, slt, T(g['dS_' + tag]), packed=packed)
            e = 0.0
            for got, key in ((d_im, 'dim_'), (d_s, 'ds_')):
                ref = g[key + tag]
                e = max(e, float(np.abs(got.cpu().numpy()[:, :, ::st] - ref).max()) / max(1e-9, float(np.abs(ref).max())))
            row.append('%s %.2e' % (m, e))
            if e > worst[m][0]:
                worst[m] = (e, '%s/%s' % (name, tag))
        print('%-20s %-4s D=%-4d %s' % (name, tag, im.shape[2], '   '.join(row)), flush=True)
print('worst per mode:', {m: '%.2e (%s)' % worst[m] for m in MODES})

B = 256
im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234, ragged=False)
a, b = T(im), T(s)
ilt, slt = ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev)
loss, S, (im_c, s_c, geom, buf, dS, ws, offs) = ops._triplet_forward(a, b, ilt, slt, 0.2)
one = torch.ones((), device=dev)


def rows_us(iters=200):
    pk = ops._packed_from_buf(buf, offs)
    for _ in range(20):
        ops._triplet_backward(im_c, s_c, ilt, slt, geom, pk, dS, ws, one)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops._triplet_backward(im_c, s_c, ilt, slt, geom, pk, dS, ws, one)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rep in range(3):
    out = []
    for m in MODES:
        setmode(m)
        out.append('%s %.1f us' % (m, rows_us()))
    print('B = 256 row kernel (call to call, incl. launch):', '   '.join(out), flush=True)

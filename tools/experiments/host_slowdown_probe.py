#!/usr/bin/env python3
"""Round 6: the bs-32 loss heads measured inside bench.py's long process were ~50 % slower than in a fresh process.  What in the process does that?
Times benchlib.loss_heads_bs32 (a) fresh, (b) after capturing a HIP graph of the B = 256 step, (c) after a second capture + delete,
(d) after the configs[2] retrieval calls, (e) after gc.freeze()."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch

import benchlib
from aladin_amd import ops, synth
from aladin_amd.loss import AlignmentContrastiveLoss

dev = torch.device('cuda:0')


def heads(tag):
    d = benchlib.loss_heads_bs32(dev)
    print(tag, {k: d[k] for k in ('eager_ms', 'graphed_step_ms', 'model_eager_ms', 'model_flag_ms')}, flush=True)


heads('a fresh          ')
im_np, s_np, il, sl = synth.alignment_batch(256, 34, 50, 768, seed=1234, ragged=False)
im = torch.from_numpy(im_np).to(dev).requires_grad_(True)
s = torch.from_numpy(s_np).to(dev).requires_grad_(True)
crit = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')
seed = torch.ones((), device=dev)
for _ in range(3):
    crit(im, s, il, sl).backward(gradient=seed)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    crit(im, s, il, sl).backward(gradient=seed)
for _ in range(2000):
    g1.replay()
torch.cuda.synchronize()
heads('b after 1 capture ')
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    crit(im, s, il, sl).backward(gradient=seed)
g2.replay()
torch.cuda.synchronize()
del g2
heads('c after 2nd + del ')
i_np, c_np = synth.retrieval_embeddings(5000, 768, seed=303, sigma=8.0)
a8, b8 = torch.from_numpy(i_np[0::5]).to(dev), torch.from_numpy(c_np).to(dev)
for _ in range(15):
    ops.retrieval_ranks(a8, b8)
torch.cuda.synchronize()
del a8, b8, i_np, c_np
heads('d after retrieval ')
gc.collect()
gc.freeze()
heads('e after gc.freeze ')

#!/usr/bin/env python3
"""Round 6: where does the host time of ALADModel(graphed=True) go at bs 32?  Wall clock per step of
  A  GraphedLossStep, no logger          B  GraphedLossStep + logger (deferred copies)      C  ... log='sync'
  D  ALADModel(graphed=True)(...)        E  D + str(model.logger) every step               F  eager ALADModel"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from aladin_amd import synth
from aladin_amd.alad_model import ALADModel
from aladin_amd.evaluation import LogCollector
from aladin_amd.graphs import GraphedLossStep

dev = torch.device('cuda:0')
B, R, Tn = 32, 51, 38
im, s, il, sl = synth.alignment_batch(B, R, Tn, 768, seed=7, ragged=True)
gi, gc = synth.global_embeddings(B, 768, seed=8)
x = torch.from_numpy(gi).to(dev).requires_grad_(True)
y = torch.from_numpy(gc).to(dev).requires_grad_(True)
a_s = torch.from_numpy(im).to(dev).permute(1, 0, 2).contiguous().requires_grad_(True)
b_s = torch.from_numpy(s).to(dev).permute(1, 0, 2).contiguous().requires_grad_(True)
seed = torch.ones((), device=dev)
cfg = {'training': {'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'margin': 0.2, 'measure': 'dot',
                    'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}


def wall(fn, iters=300, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def zero():
    for t in (a_s, b_s, x, y):
        t.grad = None


def step_of(gstep):
    def run():
        zero()
        loss, _ = gstep(x, y, a_s, b_s, il, sl, epoch=5)
        loss.backward(gradient=seed)
    return run


def model_of(m, read):
    m.forward_emb = lambda p, q: (x, y, a_s, b_s, il, sl, 0)

    def run():
        zero()
        loss, _ = m(None, None, epoch=5, distill_epoch=2)
        loss.backward(gradient=seed)
        if read:
            str(m.logger)
    return run


res = {}
for rep in range(2):
    m0 = ALADModel(cfg)
    res['A gstep, no logger'] = wall(step_of(GraphedLossStep(m0)))
    m1 = ALADModel(cfg); m1.logger = LogCollector()
    g1 = GraphedLossStep(m1)
    res['B gstep + logger deferred'] = wall(step_of(g1)); g1.flush()
    m2 = ALADModel(cfg); m2.logger = LogCollector()
    res['C gstep + logger sync'] = wall(step_of(GraphedLossStep(m2, log='sync')))
    m3 = ALADModel(cfg, graphed=True); m3.logger = LogCollector()
    res['D model flag'] = wall(model_of(m3, False))
    m4 = ALADModel(cfg, graphed=True); m4.logger = LogCollector()
    res['E model flag + logger read every step'] = wall(model_of(m4, True))
    m5 = ALADModel(cfg, graphed=False); m5.logger = LogCollector()
    res['F model eager'] = wall(model_of(m5, False), iters=100)
    m6 = ALADModel(cfg, graphed=True)
    res['G model flag, no logger'] = wall(model_of(m6, False))
    print({k: round(v, 4) for k, v in res.items()}, flush=True)

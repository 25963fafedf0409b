#!/usr/bin/env python3
"""The alignment triplet step with the sum-of-violations hinge (max_violation=False: dS dense, every pair differentiated;
no shipped config) at B = 64 / 256, next to the hardest-negative hinge.  Prints one JSON line per batch size."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from aladin_amd import synth
from aladin_amd.loss import AlignmentContrastiveLoss


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device('cuda:0')
    only = os.environ.get('DENSE_ONLY')                     # profiling: one mode, B = 256 only
    for B in ((256,) if only else (64, 128, 256)):
        im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234)
        a = torch.from_numpy(im).to(dev).requires_grad_(True)
        b = torch.from_numpy(s).to(dev).requires_grad_(True)
        out = {'batch': B}
        if only:
            from aladin_amd import ops
            ops.DENSE_BACKWARD = only == '1'
        for mv in ((False,) if only else (True, False)):
            crit = AlignmentContrastiveLoss(0.2, 'dot', mv, 'MrSw')

            def step():
                a.grad = None
                b.grad = None
                crit(a, b, il, sl).backward()
            out['max_violation_%s_ms' % mv] = round(timed(step, 30), 4)
            if not mv and not only:
                from aladin_amd import ops
                ops.DENSE_BACKWARD = False
                out['max_violation_False_per_pair_ms'] = round(timed(step, 30), 4)
                ops.DENSE_BACKWARD = True
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Where the time of the COCO-1k alignment grid goes: wall time per call, host profile (cProfile), GPU time (events).
  --img-range LO HI    image lengths (default 12 34, the bench fixture; 18 51 = images of up to 50 boxes + the global slot)
  --x-bounds a,b,...   override ops.X_CLASS_BOUNDS (A/B of the planner's image classes)
  --no-host-profile"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aladin_amd import evaluation as E, ops, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img-range', type=int, nargs=2, default=(12, 34))
    ap.add_argument('--x-bounds', type=str, default=None)
    ap.add_argument('--no-host-profile', action='store_true')
    a = ap.parse_args()
    if a.x_bounds:
        ops.X_CLASS_BOUNDS = tuple(int(v) for v in a.x_bounds.split(','))
    dev = torch.device('cuda:0')
    n = 1000
    images, captions, il, cl = synth.eval_sets(n, 768, seed=9, img_len_range=tuple(a.img_range))
    print('image lengths %d..%d, X_CLASS_BOUNDS %s' % (min(il), max(il), ops.X_CLASS_BOUNDS), flush=True)
    ia = torch.from_numpy(images[0::5]).to(dev)
    ca = torch.from_numpy(captions).to(dev)
    ilen = il[0::5]
    for prec in ('fp16', 'split'):
        ops.set_eval_precision(prec)
        for bucket in (True, False):
            saved = ops.bucket_plan
            if not bucket:
                ops.bucket_plan = lambda *a: None
            ops._PLAN_CACHE.clear()
            fn = lambda: E.compute_sim_matrix(ia, ca, ilen, cl, mode='alignment')
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            t_issue = (time.perf_counter() - t0) / 5 * 1e3
            torch.cuda.synchronize()
            t_wall = (time.perf_counter() - t0) / 5 * 1e3
            print('precision %-5s bucketed %-5s  wall %.2f ms/call  host issue %.2f ms/call  gpu (events) %.2f ms/call'
                  % (prec, bucket, t_wall, t_issue, e0.elapsed_time(e1) / 5), flush=True)
            if bucket and prec == 'fp16' and not a.no_host_profile:
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                pr.disable()
                pstats.Stats(pr).sort_stats('tottime').print_stats(14)
            ops.bucket_plan = saved
            ops._PLAN_CACHE.clear()


if __name__ == '__main__':
    main()

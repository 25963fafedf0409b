#!/usr/bin/env python3
"""One-off CPU baseline at the headline size (SURVEY.md 8(d) "CPU timing plan"): the reference's own dataflow
(expand + batched matmul + masks, oracle/faithful_torch.py = alad/loss.py:79-159 op for op) at B=256, R=34,
T=50, D=768, forward and forward+backward, on the GPU box's host cores with a thread sweep {8, 32, nproc}.
Needs ~40 GB of host memory and minutes of CPU time, so it is not part of the default bench.py run; its
result is committed as profiles/r02_cpu_baseline_b256.json and attached to bench.py's cpu_baseline as `b256`.

    python tools/cpu_baseline_sweep.py [--batch 256] [--out gpurun_out/r02_cpu_baseline_b256.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch

import faithful_torch as FT
from aladin_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r02_cpu_baseline_b256.json'))
    ap.add_argument('--threads', default='8,32,all')
    args = ap.parse_args()
    nproc = os.cpu_count() or 1
    threads = sorted({min(nproc, nproc if t == 'all' else int(t)) for t in args.threads.split(',')})
    Bt = args.batch
    im, s, il, sl = synth.alignment_batch(Bt, 34, 50, 768, seed=1234, ragged=False)
    a, b = torch.from_numpy(im), torch.from_numpy(s)
    res = {}
    for th in threads:
        torch.set_num_threads(th)
        r = {}
        for tag, bwd in (('fwd', False), ('fwd_bwd', True)):
            t0 = time.perf_counter()
            FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=bwd)
            first = time.perf_counter() - t0
            if first < 90.0:                      # a second, warm repetition when it is affordable
                t0 = time.perf_counter()
                FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=bwd)
                r[tag] = min(first, time.perf_counter() - t0)
            else:
                r[tag] = first
            print('threads %d %s: %.2f s/step' % (th, tag, r[tag]), flush=True)
        res[th] = r
    best = min(res, key=lambda k: res[k]['fwd_bwd'])
    out = {'workload': 'reference dataflow (oracle/faithful_torch.py) B=%d R=34 T=50 D=768 fp32' % Bt, 'host_cores': nproc,
           'best_threads': best, 'pairs_per_s_fwd_bwd': round(Bt * Bt / res[best]['fwd_bwd'], 1),
           'pairs_per_s_fwd': round(Bt * Bt / min(v['fwd'] for v in res.values()), 1),
           's_per_step': {str(k): {t: round(v, 3) for t, v in r.items()} for k, r in res.items()}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

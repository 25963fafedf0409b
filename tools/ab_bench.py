#!/usr/bin/env python3
"""A/B (or A/B/C ...) of whole library builds on ONE box, interleaved: boxes differ by up to 10 % and the
chip's clock drifts, so kernel variants are only comparable when alternated in one session.

    make -C aladin_amd/csrc && cp aladin_amd/lib/libaladin_hip.so /tmp/a.so      # variant A
    ... edit a kernel ... make ... cp ... /tmp/b.so                                 # variant B
    gpurun -- python tools/ab_bench.py --lib a=aladin_amd/lib/a.so --lib b=aladin_amd/lib/b.so --reps 3

(the .so files must live inside the repo to travel to the GPU box).  Each run is `bench.py --no-cpu-baseline`
in a child process with ALADIN_LIB set; prints ms/step, the score kernel's event-timed duration and the loss
(which must agree between variants)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', action='append', required=True,
                    help='name=path/to/lib.so[;ENV=VALUE;...] (repeatable; the ENV settings select a variant inside the diag build)')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--steps', type=int, default=1000)
    args = ap.parse_args()
    libs = []
    for a in args.lib:
        name, rest = a.split('=', 1)
        parts = rest.split(';')
        libs.append((name, parts[0], dict(p.split('=', 1) for p in parts[1:] if p)))
    rows = {n: [] for n, _, _ in libs}
    for _ in range(args.reps):
        for name, path, extra in libs:
            env = dict(os.environ, ALADIN_LIB=os.path.abspath(path), **extra)
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', str(args.steps), '--warmup',
                                  str(max(10, args.steps // 10)), '--no-cpu-baseline', '--no-eval', '--repeats', '3'], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith('{')]
            if not line:
                print(name, 'FAILED', out.stderr[-300:])
                continue
            d = json.loads(line[-1])
            rows[name].append((d['ms_per_step'], d['roofline']['kernel_us'], d['config']['loss']))
            print('%-8s step %.4f ms  score kernel %.2f us  loss %.6f' % (name, *rows[name][-1]), flush=True)
    for name, r in rows.items():
        if r:
            print('%-8s mean step %.4f ms  mean kernel %.2f us over %d runs' % (name, sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r), len(r)))


if __name__ == '__main__':
    main()

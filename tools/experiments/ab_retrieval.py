#!/usr/bin/env python3
"""Interleaved A/B of library builds on the configs[2] retrieval (tools/bench_retrieval.py in a child process per run, ALADIN_LIB set):
    gpurun -- python tools/experiments/ab_retrieval.py --lib a=aladin_amd/lib/ab_a.so --lib b=aladin_amd/lib/ab_b.so --reps 3 --only sigma
prints fused ms per data set and variant (every run checks the ranks against the two-step path)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--lib', action='append', required=True)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--only', default='sigma')
args = ap.parse_args()
libs = [a.split('=', 1) for a in args.lib]
rows = {}
for rep in range(args.reps):
    for name, path in libs:
        env = dict(os.environ, ALADIN_LIB=os.path.abspath(os.path.join(ROOT, path)))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_retrieval.py'), '--only', args.only], env=env, capture_output=True, text=True)
        lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith('{')]
        if not lines:
            print(name, 'FAILED', out.stderr[-400:])
            continue
        for d in lines:
            rows.setdefault((d['data'][:40], name), []).append(d['fused_ms'])
        print(rep, name, [(d['data'][-22:], d['fused_ms']) for d in lines], flush=True)
for (data, name), v in sorted(rows.items()):
    print('%-42s %-6s mean %.4f ms  min %.4f  (%d runs)' % (data, name, sum(v) / len(v), min(v), len(v)))

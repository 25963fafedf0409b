#!/usr/bin/env python3
"""Round 6: BASELINE configs[3] at its own size -- 8 ranks x B = 256, global 2048 x 2048 -- executed END TO END through bench.py on ONE GPU
(`--gpus 8 --shared-gpu --selftest-batch 256`: real kernels, gloo exchange, the ranks time-slice the GPU: no performance meaning), and the
global loss checked against the single-device step on the concatenated 2048-sample batch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--shared-gpu', '--selftest-batch', '256', '--steps', '3', '--warmup', '1'],
                   capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
print('rc', p.returncode, p.stderr[-800:] if p.returncode else '')
d = json.loads([l for l in p.stdout.splitlines() if l.strip().startswith('{')][-1])
c = d['config']
print(json.dumps({k: c[k] for k in ('workload', 'global_pairs_per_step', 'loss', 'collectives', 'bwd_exchange', 'bwd_exchange_tuning_ms', 'bwd_partners', 'phases_ms')}))
import numpy as np
import torch
from aladin_amd import synth
from aladin_amd.loss import AlignmentContrastiveLoss
parts = [synth.alignment_batch(256, 34, 50, 768, seed=1234 + 17 * r, ragged=False) for r in range(8)]
dev = torch.device('cuda:0')
im = torch.from_numpy(np.concatenate([q[0] for q in parts])).to(dev)
s = torch.from_numpy(np.concatenate([q[1] for q in parts])).to(dev)
loss = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(im, s, sum((q[2] for q in parts), []), sum((q[3] for q in parts), []))
print('single-device loss of the concatenated 2048-sample batch: %r   sharded (8 ranks): %r   equal: %s' % (float(loss), c['loss'], float(loss) == c['loss']))

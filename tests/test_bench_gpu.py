"""GPU tier: the driver-facing script itself.  `bench.py` in a child process (short protocol), default and --force-sharded
(the multi-GPU step -- RCCL collectives, both backward exchanges timed -- on one rank): one JSON line, LAST on stdout, with the
contract's fields; the sharded step reproduces the single-device loss."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline')


def _bench(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--warmup', '5', '--repeats', '2',
                        '--preroll-s', '0.3', '--no-cpu-baseline', '--no-eval'] + list(extra),
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                          # the driver reads the last line
    for k in REQUIRED:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 40 and d['warmup'] == 5 and d['unit'] == 'pairs/s' and d['dtype'] == 'f16'
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and d['scaling'] == 'weak' and d['data'] == 'synthetic'
    assert abs(d['value'] - 65536 / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0.3 < r['frac'] < 0.8
    assert 'workload' in d['config'] and 'model' not in d['config']
    return d


def test_bench_single_gpu_line():
    d = _bench()
    assert d['config']['launch'] in ('hipgraph', 'eager') and d['config']['bwd_partners'] == 'exact'
    assert 0.1 < d['ms_per_step'] < 0.5


def test_bench_sharded_step_on_one_rank():
    d0 = _bench()
    d = _bench('--force-sharded')
    c = d['config']
    assert c['collectives'] == {'backend': 'nccl', 'ranks': 1, 'launcher': 'external'}
    assert c['bwd_exchange'] in ('dense', 'sparse') and set(c['bwd_exchange_tuning_ms']) == {'dense', 'sparse'}
    assert all(v is not None and v > 0 for v in c['bwd_exchange_tuning_ms'].values())          # both exchanges ran
    for phase in ('pack+issue_gathers', 'local_block', 'S_allgather', 'hinge', 'bwd_start'):
        assert phase in c['phases_ms'], phase
    assert c['loss'] == d0['config']['loss']                           # the global-batch loss of one rank IS the single-device loss

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
    assert d['config']['launch'] in ('hipgraph', 'eager') and d['config']['bwd_partners'] == 'fp16'
    lib = d['config']['library']                     # the timed binary is the tree's sources (link-time stamp, tools/srchash.py)
    assert lib['current'] is True and lib['built_from'] == lib['tree'], lib
    assert 0.1 < d['ms_per_step'] < 0.5
    # the precision trade behind the headline is in the line: the same step with the exact row step, timed the same way
    x = d['config']['bwd_exact_ms_per_step']
    assert isinstance(x, float) and 0.9 * d['ms_per_step'] < x < 0.6, x


def test_bench_sharded_step_on_one_rank():
    d0 = _bench()
    d = _bench('--force-sharded')
    c = d['config']
    assert c['collectives'] == {'backend': 'nccl', 'ranks': 1, 'launcher': 'external'}
    assert c['bwd_partners'] == 'fp16' and c['bwd_exact_ms_per_step'] is None
    assert c['bwd_exchange'] in ('dense', 'sparse') and set(c['bwd_exchange_tuning_ms']) == {'dense', 'sparse'}
    assert all(v is not None and v > 0 for v in c['bwd_exchange_tuning_ms'].values())          # both exchanges ran
    for phase in ('pack+issue_gathers', 'local_block', 'S_allgather', 'hinge', 'bwd_start'):
        assert phase in c['phases_ms'], phase
    assert c['loss'] == d0['config']['loss']                           # the global-batch loss of one rank IS the single-device loss


def test_bench_four_ranks_sharing_this_gpu_run_the_real_sharded_step():
    """`bench.py --gpus 4 --shared-gpu`: the driver-facing script's multi-rank path with the REAL kernels on a one-GPU box -- self-launch
    (aladin_amd.launch), four ranks on cuda:0, gloo for the exchange (RCCL refuses two ranks per device), the real step loop, exchange
    tuning (dense reduce-scatter AND pair-driven all-to-all), PhaseRecorder, watchdog, ONE JSON line.  The global loss must be the
    single-device loss of the concatenated 256-sample batch, bit for bit (every rank scores its caption block with the same kernels)."""
    import numpy as np
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--shared-gpu', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    c = d['config']
    assert d['n_gpus'] == 4 and d['data'] == 'shared-gpu-selftest' and d['value'] == 0.0 and 'SHARED-GPU SELF-TEST' in c['workload']
    assert c['collectives'] == {'backend': 'gloo', 'ranks': 4, 'launcher': 'self (aladin_amd.launch)'}
    assert set(c['bwd_exchange_tuning_ms']) == {'dense', 'sparse'} and all(v for v in c['bwd_exchange_tuning_ms'].values())
    assert c['bwd_partners'] == 'fp16' and c['library']['current'] is True
    for phase in ('pack+issue_gathers', 'local_block', 'gather_wait', 'remote_rows', 'S_allgather', 'hinge', 'bwd_start'):
        assert phase in c['phases_ms'], phase
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    parts = [synth.alignment_batch(64, 34, 50, 768, seed=1234 + 17 * r, ragged=False) for r in range(4)]
    dev = torch.device('cuda:0')
    im = torch.from_numpy(np.concatenate([q[0] for q in parts])).to(dev)
    s = torch.from_numpy(np.concatenate([q[1] for q in parts])).to(dev)
    loss = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(im, s, sum((q[2] for q in parts), []), sum((q[3] for q in parts), []))
    assert c['loss'] == float(loss)


def _bench_raw(*args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--repeats', '1', '--preroll-s', '0.3'] + list(args),
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])


def test_bench_secondary_fields_are_in_the_line():
    """what DESIGN.md claims next to the headline is timed by the driver's own run -- configs[2] retrieval with its
    data-dependent cost, the shipped data shape, the loss heads at the shipped batch size, configs[4] end to end, the COCO-1k
    alignment grid -- and the roofline carries the kernel, forward-chain and step fractions side by side."""
    d = _bench_raw('--no-cpu-baseline')
    c, r = d['config'], d['roofline']
    for k in ('eval_config3', 'shipped_shape', 'loss_heads_bs32', 'e2e_config4', 'alignment_retrieval_coco1k'):
        assert k in c and 'error' not in c[k], (k, c.get(k))
    e = c['eval_config3']
    # `ms` is timed on the SURVEY 8(d) input (Recall@1 of both directions in 40-80 %) and must run on the screen, not the exact path.
    # Which tiles skip their analysis depends on scheduling (include/aladin_hip.h): the statistics are held to their invariants and the
    # timings to wide bounds only -- the outputs themselves are pinned int for int by tests/test_gpu_parity.py
    assert 0.1 < e['ms'] < 0.6 and 0.1 < e['all_exact_ms'] < 1.5 and len(e['by_data']) == 3          # (10-call timings on a shared box: gross bounds only)
    assert 40.0 <= e['R@1_i2t'] <= 80.0 and 40.0 <= e['R@1_t2i'] <= 80.0 and e['rescored_pairs'] <= e['listed_pairs']
    assert 0 <= e['exact_tiles'] <= e['tiles'] and all(0.1 < b['ms'] < 1.5 for b in e['by_data'])
    assert all(0 <= b['exact_tiles'] <= e['tiles'] and b['rescored_pairs'] <= b['listed_pairs'] for b in e['by_data'])
    assert 0.05 < e['frac'] < 0.6 and e['screen_kernel_us'] > 50
    assert 0.1 < c['shipped_shape']['ms_per_step'] < 0.6 and 0.2 < c['shipped_shape']['score_kernel_frac'] < 0.8
    h = c['loss_heads_bs32']
    assert 0 < h['graph_replay_only_ms'] < 1.0 and 0 < h['graphed_step_ms'] < 1.0 and h['eager_ms'] > 0
    assert 0 < h['model_flag_ms'] < 1.0 and 0 < h['model_eager_ms'] < 2.0 and h['model_flag_logged_ms'] > 0          # ALADModel(graphed=True): the unchanged train loop gets the replay
    x = c['e2e_config4']
    assert x['fp32_step_ms'] > x['loss_heads_ms'] > 0 and 0 < x['loss_heads_share_of_fp32_step'] < 0.1 and x['bf16_autocast_batched_passes_step_ms'] > 0
    a = c['alignment_retrieval_coco1k']
    assert 0 < a['fp16_ms'] < a['split_ms'] < 100
    assert 0.2 < r['step_frac'] < r['forward_chain_frac'] < r['frac'] < 0.8
    assert r['traffic'] is None or r['traffic'] > 1e8                   # null when the committed PMC summary belongs to other kernel sources
    assert (r['traffic'] is None) == str(r['traffic_source']).startswith('stale')


def test_bench_cpu_baseline_is_live():
    """cpu_baseline: BASELINE configs[0] (B = 16) swept over thread counts, and the B = 256 step of the same dataflow timed in THIS run."""
    d = _bench_raw('--no-eval', timeout=1500)
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['unit'] == 'pairs/s' and cb['value'] > 0 and cb['cores'] >= 1 and 'sample' in cb
    b = cb['b256']
    assert b.get('source') == 'live' or str(b.get('source', '')).startswith('committed'), b
    if b.get('source') == 'live':
        assert b['steps_timed'] >= 1 and 1e3 < b['pairs_per_s_fwd_bwd'] < 1e6

"""CPU tier: `bench.py --gpus N` starts its own ranks (aladin_amd/launch.py).  The reference has no launcher
(single process, alad/train.py:251-255); BASELINE configs[3] needs one.  A stub rank body under gloo checks what the
launcher owes the driver: N ranks with the torch.distributed.run environment on 127.0.0.1, the JSON line LAST on
stdout, the children's failure as a non-zero return code, no launch when already inside a rank."""
import io
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

STUB = os.path.join(ROOT, 'tests', 'helpers', 'launch_stub.py')


def _run(argv, nproc, **kw):
    from aladin_amd import launch
    out, err = io.StringIO(), io.StringIO()
    env = dict(os.environ, OMP_NUM_THREADS='1')
    rc, line = launch.run_ranks(STUB, argv, nproc, env=env, out=out, err=err, **kw)
    return rc, line, out.getvalue(), err.getvalue()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('nproc', [2, 3])
def test_launcher_starts_n_ranks_and_keeps_the_result_line_last(nproc):
    rc, line, out, err = _run(['--gpus', str(nproc)], nproc)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert lines[-1] == line                                           # the driver reads the last line
    res = json.loads(line)
    assert res['n_gpus'] == nproc and res['asked'] == nproc
    assert res['local_ranks_plus_1'] == list(range(1, nproc + 1))      # every rank took part, LOCAL_RANK = rank on one node
    assert res['master'][0] == '127.0.0.1' and int(res['master'][1]) > 0
    assert res['self_launched'] == '1'
    assert 'banner: not the result line' in out and out.count('late noise') == nproc      # nothing swallowed
    assert '{"looks": "like json but is followed by the real line"}' in lines[:-1]


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch():
    rc, line, out, err = _run(['--gpus', '2', '--fail-rank', '1'], 2)
    assert rc != 0
    assert 'failing on purpose' in err


@pytest.mark.timeout(300)
def test_a_failed_run_withholds_its_result_line_from_stdout():
    """rank 0 prints its JSON line, then another rank dies: the driver reads the LAST stdout line and must not find it."""
    rc, line, out, err = _run(['--gpus', '2', '--fail-after-result'], 2)
    assert rc != 0 and line is None
    assert '"n_gpus"' not in out
    assert 'withheld from stdout' in err and '"n_gpus": 2' in err


@pytest.mark.timeout(300)
def test_a_lost_rendezvous_port_is_retried_once(tmp_path):
    flag = str(tmp_path / 'first_attempt_done')
    rc, line, out, err = _run(['--gpus', '2', '--bind-failure-once', flag], 2)
    assert rc == 0, err
    assert 'retrying once on a fresh port' in err and json.loads(line)['n_gpus'] == 2


@pytest.mark.timeout(600)
def test_eight_ranks():
    """configs[3]'s world size under gloo: the launcher brings up 8 ranks on 127.0.0.1 and every one takes part."""
    rc, line, out, err = _run(['--gpus', '8'], 8)
    assert rc == 0, err
    res = json.loads(line)
    assert res['n_gpus'] == 8 and res['local_ranks_plus_1'] == list(range(1, 9))


@pytest.mark.timeout(300)
def test_success_without_a_result_line_is_an_error():
    rc, line, out, err = _run(['--gpus', '2', '--no-result'], 2)
    assert rc == 1 and line is None and 'without printing a result line' in err


@pytest.mark.timeout(300)
def test_timeout_ends_the_ranks():
    rc, line, out, err = _run(['--gpus', '2', '--hang'], 2, timeout=20)
    assert rc == 124 and line is None


@pytest.mark.timeout(900)
def test_bench_py_eight_ranks_stub_step():
    """the REAL bench.py -- its argument parsing, self-launch, rendezvous, barriers, MAX over ranks and JSON
    line -- at configs[3]'s world size, with a stub step on CPU (gloo).  No scaling number is claimed: this is the protocol."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--stub-step', '--steps', '5', '--warmup', '2'],
                         env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [l for l in out.stdout.splitlines() if l.strip()][-1]
    res = json.loads(last)
    assert res['n_gpus'] == 8 and res['data'] == 'stub' and res['steps'] == 5
    assert res['config']['collectives'] == {'backend': 'gloo', 'ranks': 8, 'launcher': 'self (aladin_amd.launch)'}
    assert res['config']['stub_result_ok'] is True


@pytest.mark.timeout(900)
def test_bench_py_eight_ranks_cpu_standin_runs_the_real_sharded_step():
    """the REAL step loop of bench.py at configs[3]'s world size -- the fast sharded node of aladin_amd/distributed.py,
    the exchange tuning (dense AND pair-driven sparse backward), PhaseRecorder, the stall watchdog and the one JSON line -- under gloo
    with the HIP entry points replaced by tests/helpers/cpu_standins.py (`--stub-step` above touches none of distributed.py).  The
    loss of the global 512 x 512 batch must be the unsharded composition's on the concatenated batch."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--cpu-standin', '--steps', '2', '--warmup', '1'],
                         env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    cfg = res['config']
    assert res['n_gpus'] == 8 and res['data'] == 'cpu-standin' and res['value'] == 0.0 and res['steps'] == 2
    assert cfg['collectives'] == {'backend': 'gloo', 'ranks': 8, 'launcher': 'self (aladin_amd.launch)'}
    assert set(cfg['bwd_exchange_tuning_ms']) == {'dense', 'sparse'} and all(v for v in cfg['bwd_exchange_tuning_ms'].values())
    assert cfg['bwd_exchange'] == min(cfg['bwd_exchange_tuning_ms'], key=cfg['bwd_exchange_tuning_ms'].get)
    phases = list(cfg['phases_ms'])
    assert phases[:6] == ['pack+issue_gathers', 'local_block', 'gather_wait', 'remote_rows', 'S_allgather', 'hinge'] and phases[-1] == 'autograd_tail'
    assert ('bwd_give_back' in phases) == (cfg['bwd_exchange'] == 'sparse') and ('bwd_reduce_scatter' in phases) == (cfg['bwd_exchange'] == 'dense')
    calls = cfg['standin_calls']
    assert calls['scores_from_packed'] == 8 * calls['hinge_raw'] and calls['align_backward'] == calls['hinge_raw']          # 8 rank blocks scored per step
    assert cfg['bwd_partners'] == 'fp16' and calls['pack_sets'] >= 1          # the pair-driven exchange packs its compact problem in the backward
    # the global loss = the same stand-ins composed on ONE process over the concatenated batch (rank r's batch: seed 1234 + 17 r)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    import numpy as np
    import torch
    import cpu_standins
    from aladin_amd import synth
    parts = [synth.alignment_batch(64, 34, 50, 16, seed=1234 + 17 * r, ragged=False) for r in range(8)]
    im = torch.from_numpy(np.concatenate([p[0] for p in parts]))
    s = torch.from_numpy(np.concatenate([p[1] for p in parts]))
    loss, _, _, _ = cpu_standins.single_process_step(im, s, sum((p[2] for p in parts), []), sum((p[3] for p in parts), []))
    np.testing.assert_allclose(cfg['loss'], float(loss), rtol=1e-5)


def test_no_self_launch_inside_a_rank_or_for_one_gpu():
    from aladin_amd import launch
    assert launch.needs_self_launch(8, {})
    assert not launch.needs_self_launch(1, {})
    assert not launch.needs_self_launch(8, {'WORLD_SIZE': '8', 'RANK': '3', 'LOCAL_RANK': '3'})
    cmd = launch.rank_command('bench.py', ['--gpus', '8', '--steps', '5'], 8, 29999, python='python3')
    assert cmd == ['python3', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                   '--master-port', '29999', 'bench.py', '--gpus', '8', '--steps', '5']


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_its_ranks_here_and_reports_their_failure():
    """No GPU in this container: both ranks of `python bench.py --gpus 2` must START (the old bench refused to) and fail
    with bench's own 'needs an MI355X' message; the launcher hands their failure on."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=280, env=dict(os.environ, OMP_NUM_THREADS='1'))
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU box: covered by the gpu tier')
    assert p.returncode != 0
    assert 'needs an MI355X' in p.stderr and 'must be launched with' not in p.stderr

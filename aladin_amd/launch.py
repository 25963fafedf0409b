"""Single-node rank launcher for `bench.py --gpus N` (BASELINE configs[3]).

The reference is single-process, single-device (alad/train.py:251-255): it has nothing to launch.  This
framework runs one process per GPU, so a plain `python bench.py --gpus N` has to START its N ranks itself.
It does so as CHILD processes (`python -m torch.distributed.run`, rendezvous on 127.0.0.1) before this process
has imported torch or touched the GPU -- a process that has initialised the HIP runtime must never exec another
program on this pool -- relays the children's output, keeps the result (JSON) line LAST on stdout and exits with
the children's return code.  Standard library only: importing this module starts no runtime.
"""
import json
import os
import socket
import subprocess
import sys
import threading
import time

RANK_ENV = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')


def needs_self_launch(gpus, environ=None):
    """True when N > 1 ranks were asked for and this process is not already one of them."""
    environ = os.environ if environ is None else environ
    return int(gpus) > 1 and not any(k in environ for k in RANK_ENV)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def rank_command(script, argv, nproc, port, python=None):
    """The command the driver itself uses for N > 1 (one rank per GPU of ONE node)."""
    return [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nproc)),
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), script] + list(argv)


def _is_result_line(line):
    line = line.strip()
    if not (line.startswith('{') and line.endswith('}')):
        return False
    try:
        return isinstance(json.loads(line), dict)
    except ValueError:
        return False


# what torch.distributed.run / the TCPStore print when the rendezvous port was taken between free_port() and their bind.  The retry
# needs the rendezvous message itself, a failure within RETRY_WINDOW_S of the launch and NO rank output relayed yet (a
# rank's own socket error after GPU work had started must not trigger a second full run with duplicated output)
_BIND_FAILURES = ('The server socket has failed to listen', 'failed to bind to', 'EADDRINUSE')
RETRY_WINDOW_S = 60.0


def run_ranks(script, argv, nproc, env=None, python=None, out=None, err=None, timeout=None):
    """Start `nproc` ranks of `script argv...` as children and wait for them.

    stdout of the children is relayed line by line, except that the LAST JSON-object line is held back and
    written after everything else (RCCL and the launcher print banners of their own; the driver reads the last
    line) -- and only when the ranks SUCCEEDED: the result line of a failed run goes to stderr, marked, so that a
    reader of stdout's last line cannot take a partial result for a valid one.  stderr is relayed as it comes.
    Returns (return code, result line or None): the children's return code, or 1 when they all succeeded without
    printing a result line; on `timeout` seconds the whole process group of the launcher is ended and 124 returned.
    The rendezvous port is found by bind-and-release, which another job can win before torch.distributed.run binds it:
    a launch that dies on the bind is repeated once on a fresh port."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    rc, result, err_tail, elapsed, n_out = _run_ranks_once(script, argv, nproc, env, python, out, err, timeout)
    if rc not in (0, 124) and result is None and n_out == 0 and elapsed < RETRY_WINDOW_S and any(sig in err_tail for sig in _BIND_FAILURES):
        err.write('aladin_amd.launch: rendezvous port was taken, retrying once on a fresh port\n')
        rc, result, err_tail, _, _ = _run_ranks_once(script, argv, nproc, env, python, out, err, timeout)
    if result is not None:
        if rc == 0:
            out.write(result)
            out.flush()
        else:
            err.write('aladin_amd.launch: ranks FAILED (rc %d); their result line is withheld from stdout: %s' % (rc, result))
            err.flush()
    elif rc == 0:
        err.write('aladin_amd.launch: %d ranks exited 0 without printing a result line\n' % int(nproc))
        rc = 1
    return rc, (result.strip() if result and rc == 0 else None)


def _run_ranks_once(script, argv, nproc, env, python, out, err, timeout):
    child_env = dict(os.environ if env is None else env)
    for k in RANK_ENV + ('MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK', 'LOCAL_WORLD_SIZE', 'ROLE_RANK'):
        child_env.pop(k, None)
    # the host driver only supports dmabuf IPC: RCCL / device-memory sharing across processes needs this
    child_env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    child_env['ALADIN_SELF_LAUNCHED'] = '1'
    cmd = rank_command(script, argv, nproc, free_port(), python)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=child_env, text=True, bufsize=1,
                            start_new_session=True)
    result = [None]
    err_tail = []
    n_out = [0]
    t_start = time.monotonic()

    def pump_out():
        for line in proc.stdout:
            n_out[0] += 1
            if _is_result_line(line):
                if result[0] is not None:          # an earlier candidate was ordinary output after all
                    out.write(result[0])
                result[0] = line if line.endswith('\n') else line + '\n'
            else:
                out.write(line)
            out.flush()

    def pump_err():
        for line in proc.stderr:
            err.write(line)
            err.flush()
            err_tail.append(line)
            del err_tail[:-200]

    threads = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for th in threads:
        th.start()
    try:
        rc = proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGTERM)    # exactly the session started above: launcher + its ranks
        except ProcessLookupError:
            pass
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            proc.wait()
        rc = 124
    for th in threads:
        th.join(timeout=10)
    return rc, result[0], ''.join(err_tail), time.monotonic() - t_start, n_out[0]

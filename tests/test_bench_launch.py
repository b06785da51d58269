"""CPU-only: `bench.py --gpus N` must start N ranks itself (gloo rendezvous, no RCCL) and can
never print a line whose n_gpus differs from --gpus.  Drives the real launch / barrier /
max-over-ranks / relay code of bench.py with its `--stub-cpu` stand-in stage (no HIP)."""

import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(extra, env_extra=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--stub-cpu', '--steps', '2', '--warmup', '1', *extra],
                          capture_output=True, text=True, env=env, timeout=300)


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.strip().startswith('{')]


def test_gpus2_without_launcher_starts_two_ranks():
    r = _run(['--gpus', '2'])
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1                      # exactly one line, relayed from rank 0
    out = lines[0]
    assert out['n_gpus'] == 2 and out['ranks_ran'] == 2 and len(out['per_rank_MPps']) == 2
    assert out['scaling'] == 'weak' and out['data'] == 'stub'
    # whole-job value = all ranks' frames over the slowest rank's time <= sum of the per-rank rates
    assert out['value'] <= sum(out['per_rank_MPps']) * (1 + 1e-3)


def test_gpus1_runs_in_process():
    r = _run(['--gpus', '1'])
    assert r.returncode == 0, r.stderr
    (out,) = _json_lines(r.stdout)
    assert out['n_gpus'] == 1 and out['ranks_ran'] == 1


def test_world_size_mismatch_is_an_error_not_a_silent_single_rank():
    # a launcher that started ONE rank for --gpus 2 (round-1 behaviour: printed n_gpus = 1, exit 0)
    r = _run(['--gpus', '2'], env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}, drop=())
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert '--gpus 2' in r.stderr


def test_under_an_external_launcher_two_ranks():
    """The documented driver form (torch.distributed.run sets RANK / WORLD_SIZE): ranks use the
    launcher's environment as is; rank 0 prints the line."""
    import socket

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items()}
    procs = []
    for rank in range(2):
        env = dict(base, WORLD_SIZE='2', RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / 'bench.py'), '--stub-cpu', '--gpus', '2', '--steps', '2', '--warmup', '1'],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    (line,) = _json_lines(outs[0][0])
    assert line['n_gpus'] == 2 and _json_lines(outs[1][0]) == []


def test_gpus8_rehearsal_eight_ranks():
    """The shape the driver's scaling run has (BASELINE.json configs[3]: 8 ranks of one node), rehearsed on the CPU:
    free-port choice, 8 children, gloo barrier + max-over-ranks, reaping, ONE relayed line with ranks_ran == 8."""
    r = _run(['--gpus', '8'], env_extra={'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr
    (out,) = _json_lines(r.stdout)
    assert out['n_gpus'] == 8 and out['ranks_ran'] == 8 and len(out['per_rank_MPps']) == 8
    assert out['scaling'] == 'weak' and out['value'] <= sum(out['per_rank_MPps']) * (1 + 1e-3)
    assert r.stdout.count('{') >= 1 and len(_json_lines(r.stdout)) == 1  # no rank but 0 prints a line


def test_gpus8_rehearsal_with_pinned_ranks():
    """--pin-cpus: the parent cuts the host cores it may use into one slice per rank and every child pins itself to its slice
    before it starts a thread; the line reports the cores each rank ended up with (8 ranks on this container's 8 cores: one
    each) -- rank r's launch thread cannot migrate onto another rank's core."""
    r = _run(['--gpus', '8', '--pin-cpus'], env_extra={'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr
    (out,) = _json_lines(r.stdout)
    ncores = len(os.sched_getaffinity(0))
    assert out['ranks_ran'] == 8 and out['pinned'] is True
    assert out['cpus_per_rank'] == [max(1, ncores // 8)] * 8

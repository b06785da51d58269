"""CPU-only: `bench.py --gpus N` must start N ranks itself (gloo rendezvous, no RCCL) and can
never print a line whose n_gpus differs from --gpus.  Drives the real launch / barrier /
max-over-ranks / relay code of bench.py with its `--stub-cpu` stand-in stage (no HIP)."""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

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


def _fake_sysfs(root, gpu_nodes, node_cpus):
    """A sysfs tree with one CPU topology node and len(gpu_nodes) GPUs (GPU i on PCI bus 0x10 + i, NUMA node gpu_nodes[i]; -1 or
    None = no numa_node entry) and the cpulists of the NUMA nodes."""
    topo = root / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    (topo / '0').mkdir(parents=True)
    (topo / '0' / 'properties').write_text('cpu_cores_count 64\nsimd_count 0\ndomain 0\nlocation_id 0\n')
    for i, node in enumerate(gpu_nodes):
        d = topo / str(i + 1)
        d.mkdir()
        bus = 0x10 + i
        (d / 'properties').write_text(f'cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id {bus << 8}\n')
        if node is not None:
            pci = root / 'bus' / 'pci' / 'devices' / f'0000:{bus:02x}:00.0'
            pci.mkdir(parents=True)
            (pci / 'numa_node').write_text(f'{node}\n')
    for node, cpulist in node_cpus.items():
        nd = root / 'devices' / 'system' / 'node' / f'node{node}'
        nd.mkdir(parents=True)
        (nd / 'cpulist').write_text(cpulist + '\n')


def test_numa_cpu_slices_follow_the_gpus_numa_nodes(tmp_path):
    """Each rank's cores come from ITS GPU's NUMA node (KFD topology -> PCI address -> numa_node -> cpulist, sysfs only), shared
    evenly among the ranks on that node; unknown nodes fall back to the plain index split; *_VISIBLE_DEVICES re-maps."""
    sys.path.insert(0, str(ROOT))
    import bench

    # 8 GPUs: 0-3 on node 1 (!), 4-7 on node 0 -- the crossed mapping an index split gets wrong
    _fake_sysfs(tmp_path, [1, 1, 1, 1, 0, 0, 0, 0], {0: '0-15,64-79', 1: '16-31,80-95'})
    allowed = list(range(128))
    assert bench.gpu_numa_nodes(str(tmp_path)) == [1, 1, 1, 1, 0, 0, 0, 0]
    sl = bench.numa_cpu_slices(8, str(tmp_path), allowed)
    node0, node1 = set(range(0, 16)) | set(range(64, 80)), set(range(16, 32)) | set(range(80, 96))
    assert all(set(sl[r]) <= node1 and len(sl[r]) == 8 for r in range(4)) and all(set(sl[r]) <= node0 and len(sl[r]) == 8 for r in range(4, 8))
    assert len(set().union(*map(set, sl))) == 64  # disjoint
    # fewer ranks than GPUs: rank r drives GPU r
    sl2 = bench.numa_cpu_slices(2, str(tmp_path), allowed)
    assert set(sl2[0]) | set(sl2[1]) == node1 and not (set(sl2[0]) & set(sl2[1]))
    # a restricted affinity mask (cgroup cpuset) is honoured
    sl3 = bench.numa_cpu_slices(8, str(tmp_path), list(range(0, 24)))
    assert all(set(s) <= set(range(24)) for s in sl3) and all(set(sl3[r]) <= node0 for r in range(4, 8))
    # unknown NUMA nodes -> the index split
    other = tmp_path / 'other'
    _fake_sysfs(other, [None] * 8, {})
    assert bench.numa_cpu_slices(8, str(other), allowed) == [list(range(16 * r, 16 * r + 16)) for r in range(8)]
    assert bench.numa_cpu_slices(4, str(tmp_path / 'missing'), list(range(8))) == [[0, 1], [2, 3], [4, 5], [6, 7]]
    # HIP_VISIBLE_DEVICES re-orders the devices a rank sees
    os.environ['HIP_VISIBLE_DEVICES'] = '4,5,0,1'
    try:
        assert bench.gpu_numa_nodes(str(tmp_path)) == [0, 0, 1, 1]
    finally:
        del os.environ['HIP_VISIBLE_DEVICES']


def test_ranks_under_an_external_launcher_pin_themselves_numa_aware(tmp_path):
    """Under torchrun (the driver's launch) no parent of ours hands out core slices: every rank computes its own from LOCAL_RANK /
    LOCAL_WORLD_SIZE and the (here: faked) sysfs tree, before it starts a thread.  Two ranks, GPU 0 on node 1, GPU 1 on node 0."""
    ncores = len(os.sched_getaffinity(0))
    if ncores < 4:
        pytest.skip('needs 4 cores')
    cores = sorted(os.sched_getaffinity(0))
    half = ncores // 2
    fmt = lambda cs: ','.join(map(str, cs))
    _fake_sysfs(tmp_path, [1, 0], {0: fmt(cores[:half]), 1: fmt(cores[half:2 * half])})
    import socket

    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE='2', LOCAL_WORLD_SIZE='2', RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   TDK_BENCH_SYSFS=str(tmp_path), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(ROOT / 'bench.py'), '--stub-cpu', '--gpus', '2', '--steps', '2', '--warmup', '1'],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    (line,) = _json_lines(outs[0][0])
    assert line['pinned'] is True and line['cpus_per_rank'] == [half, half]

"""CPU-only: the N > 1 path.  Two gloo ranks split a frame list, each processes its shard with
a pure-torch stand-in stage, and the union is checked on rank 0 -- the same barrier / max-over-
ranks timing protocol bench.py uses, with no collective on the data path."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_frames, out_queue):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'torch-darktable_amd'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from torch_darktable.bayer import BayerPattern, rgb_to_bayer
    from torch_darktable.sharding import shard_range

    mine = list(shard_range(num_frames, rank, world))
    dist.barrier()
    checksums = []
    for i in mine:  # independent frames: per-frame seed, no communication
        g = torch.Generator().manual_seed(1234 + i)
        frame = torch.rand(16, 16, 3, generator=g)
        checksums.append(float(rgb_to_bayer(frame, BayerPattern.RGGB).sum()))
    elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # the only collective: timing
    out_queue.put((rank, mine, checksums, float(elapsed)))
    dist.destroy_process_group()


@pytest.mark.parametrize('num_frames', [8, 5])
def test_two_rank_frame_sharding(num_frames):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted(i for _, mine, _, _ in results for i in mine)
    assert covered == list(range(num_frames))          # every frame exactly once
    assert all(abs(t - 0.002) < 1e-12 for *_, t in results)  # max over ranks reached everyone
    # the sharded result equals the single-process result
    from torch_darktable.bayer import BayerPattern, rgb_to_bayer

    by_index = {i: c for _, mine, cs, _ in results for i, c in zip(mine, cs)}
    for i in range(num_frames):
        g = torch.Generator().manual_seed(1234 + i)
        assert by_index[i] == float(rgb_to_bayer(torch.rand(16, 16, 3, generator=g), BayerPattern.RGGB).sum())


def test_shard_range_properties():
    from torch_darktable.sharding import shard_range

    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_frame_streams_single_stream_is_a_plain_loop():
    """FrameStreams with one stream touches no HIP API: the frames go through one chain object in order."""
    from torch_darktable.sharding import FrameStreams

    made = []

    def make():
        made.append(1)
        return lambda f: f * 2

    runner = FrameStreams(None, make, streams=1)
    assert runner.run([1, 2, 3]) == [2, 4, 6] and len(made) == 1
    import pytest

    with pytest.raises(ValueError):
        FrameStreams(None, make, streams=0)

"""Frame-list sharding for multi-GPU runs: frames are independent, so a batch is split into
contiguous chunks, one per rank (one process per GPU), with no collective on the data path."""

from __future__ import annotations


def shard_range(num_frames: int, rank: int, world_size: int) -> range:
    """Indices of the frames rank `rank` processes.  Contiguous chunks; the first
    `num_frames % world_size` ranks get one extra frame."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, extra = divmod(num_frames, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_frames(frames: list, rank: int, world_size: int) -> list:
    return [frames[i] for i in shard_range(len(frames), rank, world_size)]

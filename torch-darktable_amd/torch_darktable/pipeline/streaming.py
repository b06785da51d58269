"""Raw files -> device-resident byte buffers, with the disk read and the PCIe copy of the NEXT
frames overlapped with the processing of the current one.

The reference loads one file at a time on the caller's thread (`camera_settings.load_raw_bytes`:
read, `torch.frombuffer`, `.to(device)`), so a 12 MP frame pays its 19 MB upload (≈0.3 ms on PCIe
Gen5, more from pageable memory) and its file read in series with ≈0.9 ms of kernels.  Here reader
threads (a 19 MB read out of the page cache is itself ≈2 ms on one core) fill a small pool of PINNED host buffers and issues each upload on a dedicated copy stream;
the consumer's stream only waits on the event of the frame it is about to use.  No kernel of the
library is involved: this is host-side plumbing around the same `uint8` tensors `ImageProcessor`
takes.
"""

from __future__ import annotations

import queue
import sys
import threading
from collections.abc import Iterable, Iterator
from pathlib import Path

import numpy as np
import torch


class RawFrameStream:
    """Iterate over raw frames, in order, as `uint8` device tensors; reads and uploads run up to `depth`
    frames ahead on `readers` threads.

    sources : file paths, `bytes`-like objects or 1-D uint8 CPU tensors, all `frame_bytes` long
              (a `CameraSettings.bytes`-sized file including its trailing padding).
    The yielded tensor is safe to use on the stream that is current when it is yielded; it is a
    fresh allocation, so holding on to it never blocks the stream.
    """

    def __init__(self, sources: Iterable[Path | str | bytes | torch.Tensor], device: torch.device, frame_bytes: int, depth: int = 4,
                 readers: int = 2):
        if depth < 1 or readers < 1:
            raise ValueError('depth and readers must be >= 1')
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError('RawFrameStream uploads to a GPU; use load_raw_bytes for host tensors')
        self.frame_bytes = int(frame_bytes)
        self.depth = int(depth)
        self.readers = min(int(readers), self.depth)  # a reader needs a pinned slot of its own to be useful
        self._sources = iter(sources)

    def _read_into(self, src, pinned: torch.Tensor) -> None:
        """Fill one pinned slot: files are read straight into it (no intermediate copy)."""
        view = pinned.numpy()
        if isinstance(src, torch.Tensor):
            if src.numel() != self.frame_bytes or src.dtype != torch.uint8:
                raise ValueError(f'frame tensor: {src.numel()} x {src.dtype}, expected {self.frame_bytes} x torch.uint8')
            pinned.copy_(src.reshape(-1))
        elif isinstance(src, (bytes, bytearray, memoryview)):
            if len(src) != self.frame_bytes:
                raise ValueError(f'frame buffer has {len(src)} bytes, expected {self.frame_bytes}')
            view[:] = np.frombuffer(src, dtype=np.uint8)
        else:
            path = Path(src)
            size = path.stat().st_size
            if size != self.frame_bytes:
                raise ValueError(f'{path}: {size} bytes, expected {self.frame_bytes}')
            with open(path, 'rb') as f:
                got = f.readinto(view)
            if got != self.frame_bytes:
                raise OSError(f'{path}: short read ({got} of {self.frame_bytes} bytes)')

    def __iter__(self) -> Iterator[torch.Tensor]:
        dev = self.device
        copy_stream = torch.cuda.Stream(dev)
        free_slots: queue.Queue = queue.Queue()  # (pinned buffer, event of the last upload out of it)
        for _ in range(self.depth):
            free_slots.put((torch.empty(self.frame_bytes, dtype=torch.uint8).pin_memory(), None))
        source_lock = threading.Lock()
        numbered = enumerate(self._sources)
        cond = threading.Condition()
        results: dict[int, object] = {}
        state = {'next_out': 0, 'total': None, 'stop': False}

        def reader() -> None:
            with torch.cuda.device(dev):
                while True:
                    with source_lock:
                        if state['total'] is not None or state['stop']:
                            return
                        try:
                            i, src = next(numbered)
                        except StopIteration:
                            with cond:
                                state['total'] = state.get('issued', 0)
                                cond.notify_all()
                            return
                        state['issued'] = i + 1
                    try:
                        with cond:  # stay at most `depth` frames ahead of the consumer (bounds device memory)
                            cond.wait_for(lambda: i < state['next_out'] + self.depth or state['stop'])
                            if state['stop']:
                                return
                        pinned, last = free_slots.get()
                        if last is not None:
                            last.synchronize()  # the previous upload out of this slot is done: safe to overwrite
                        self._read_into(src, pinned)
                        with torch.cuda.stream(copy_stream):
                            frame = torch.empty(self.frame_bytes, dtype=torch.uint8, device=dev)
                            frame.copy_(pinned, non_blocking=True)
                            done = torch.cuda.Event()
                            done.record(copy_stream)
                        free_slots.put((pinned, done))
                        item: object = (frame, done)
                    except BaseException as e:  # noqa: BLE001  (handed to the consumer, in order)
                        item = e
                    with cond:
                        results[i] = item
                        cond.notify_all()

        workers = [threading.Thread(target=reader, name=f'raw-frame-reader-{t}', daemon=True) for t in range(self.readers)]
        # The consumer is a busy Python thread (it enqueues ~40 kernels per frame and never blocks), and a
        # reader needs the GIL for a few statements between its blocking calls: with CPython's default
        # 5 ms switch interval each hand-over can cost a frame time.  Shorten it while streaming.
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(old_interval, 2e-4))
        for t in workers:
            t.start()
        try:
            while True:
                with cond:
                    n = state['next_out']
                    cond.wait_for(lambda: n in results or (state['total'] is not None and n >= state['total']))
                    if n not in results:
                        return
                    item = results.pop(n)
                if isinstance(item, BaseException):
                    raise item
                frame, done = item
                consumer = torch.cuda.current_stream(dev)
                consumer.wait_event(done)      # device-side wait; the host does not block
                frame.record_stream(consumer)  # allocated on the copy stream, used on the consumer's
                with cond:
                    state['next_out'] = n + 1
                    cond.notify_all()
                yield frame
        finally:
            with cond:
                state['stop'] = True
                cond.notify_all()
            for t in workers:
                t.join(timeout=5.0)
            sys.setswitchinterval(old_interval)

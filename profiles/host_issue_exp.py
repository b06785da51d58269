"""Experiment: host time to issue one step of the bench (8 frames on three streams through FrameStreams) against its GPU time,
eager (one ctypes call per kernel) and as one captured HIP graph (FrameStreams.capture).
    python profiles/host_issue_exp.py > profiles/r05/host_issue.txt"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
import torch_darktable as td  # noqa: E402
from torch_darktable.sharding import FrameStreams  # noqa: E402
from torch_darktable.synthetic import synthetic_bayer  # noqa: E402

dev = torch.device('cuda', 0)
w, h, frames = 4096, 3072, 8
runner = FrameStreams(dev, lambda: bench.build_pipeline(td, dev, w, h, 'f16', 'isp')[1], streams=3)
inputs = [synthetic_bayer(h, w, seed=1234 + i, device=dev).half() for i in range(frames)]


def measure(step, label):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    # back to back: 20 steps without a sync in between (what the bench's timed region does)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{label}: host issue {sorted(t[0] for t in ts)[7] * 1e3:.3f} ms per step, issue + drain {sorted(t[1] for t in ts)[7] * 1e3:.3f} ms; '
          f'20 steps back to back: host {(t1 - t0) / 20 * 1e3:.3f} ms per step, wall {(t2 - t0) / 20 * 1e3:.3f} ms per step '
          f'= {frames * w * h / 1e6 / ((t2 - t0) / 20) :.0f} MP/s')


measure(lambda: runner.issue(inputs), 'eager, 3 streams')
cap = runner.capture(inputs)
measure(lambda: cap.replay(), 'one HIP graph per step')

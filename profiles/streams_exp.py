"""Experiment: frames of one batch spread over N HIP streams (each with its own op workspaces).

  python profiles/streams_exp.py [--streams 1 2 3] [--steps 5]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--streams', type=int, nargs='+', default=[1, 2, 3])
    ap.add_argument('--steps', type=int, default=5)
    a = ap.parse_args()
    import bench
    import torch_darktable as td
    from torch_darktable.synthetic import synthetic_bayer

    dev = torch.device('cuda', 0)
    w, h, frames = 4096, 3072, 8
    inputs = [synthetic_bayer(h, w, seed=1234 + i, device=dev).to(torch.float16) for i in range(frames)]
    for ns in a.streams:
        pipes = [bench.build_pipeline(td, dev, w, h, 'f16', 'isp')[1] for _ in range(ns)]
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]

        def step():
            for i, b in enumerate(inputs):
                with torch.cuda.stream(streams[i % ns]):
                    pipes[i % ns](b)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f'streams={ns}: {dt * 1e3:.3f} ms/step, {frames * w * h / 1e6 / dt:.0f} MP/s', flush=True)


if __name__ == '__main__':
    main()

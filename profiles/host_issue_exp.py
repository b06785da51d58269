"""Experiment: host time to issue one step of the bench (8 frames on two streams) against its GPU time -- the launches are
not the bottleneck (0.9 ms of host work per 5.0 ms step on the capture box)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd')); sys.path.insert(0, str(ROOT))
import torch, bench
import torch_darktable as td
from torch_darktable.synthetic import synthetic_bayer
dev = torch.device('cuda', 0)
w, h, frames = 4096, 3072, 8
pipes = [bench.build_pipeline(td, dev, w, h, 'f16', 'isp')[1] for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
inputs = [synthetic_bayer(h, w, seed=1234 + i, device=dev).half() for i in range(frames)]
def step():
    for i, b in enumerate(inputs):
        with torch.cuda.stream(streams[i % 2]):
            pipes[i % 2](b)
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('host issue ms per step: %.3f   total ms per step: %.3f' % (sorted(t[0] for t in ts)[5] * 1e3, sorted(t[1] for t in ts)[5] * 1e3))

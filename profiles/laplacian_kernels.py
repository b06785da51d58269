"""Per-launch device times of one Laplacian.process call at 12 MP (library event timer)."""
import sys
sys.path.insert(0, 'torch-darktable_amd'); sys.path.insert(0, '.')
import torch
import torch_darktable as td
from torch_darktable import _native
from torch_darktable.synthetic import synthetic_rgb

dev = torch.device('cuda', 0)
w, h = 4096, 3072
lum = td.compute_luminance(synthetic_rgb(h, w, seed=3, device=dev))
ws = td.Laplacian(dev, (w, h), td.LaplacianParams(6, 0.2, 1.6, 0.7, 0.3))
for _ in range(3):
    ws.process(lum)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    ws.process(lum)
b.record()
torch.cuda.synchronize()
print('Laplacian.process: %.1f us per call' % (a.elapsed_time(b) * 100))
_native.profile_enable(True)
for _ in range(5):
    ws.process(lum)
torch.cuda.synchronize()
rep = _native.profile_report()
_native.profile_enable(False)
tot = 0.0
for name, (n, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print('%-36s %3d launches/call  %8.1f us/call' % (name, n // 5, ms / 5 * 1e3))
    tot += ms / 5 * 1e3
print('sum %.1f us' % tot)

"""Measures the differences the tolerance tests bound, so that the bounds can be set to what is measured plus a margin
(tests/test_gpu_parity.py: Laplacian, fused Wiener log-luminance chain; __graft_entry__.smoke: bilateral stage).
  python profiles/bounds_probe.py > profiles/r03/bounds_probe.json     (on the GPU box)"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / 'torch-darktable_amd', ROOT / 'oracle'):
    sys.path.insert(0, str(p))
import tdk_oracle as O  # noqa: E402
import torch_darktable as td  # noqa: E402
from torch_darktable.synthetic import synthetic_rgb  # noqa: E402

dev = torch.device('cuda', 0)
gpu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
npy = lambda t: t.detach().cpu().numpy()
scene = lambda h, w, seed: synthetic_rgb(h, w, seed, 'cpu', 0.02).numpy()


def half_ulp(v):
    return 2.0 ** (np.floor(np.log2(np.maximum(np.abs(v), 2.0 ** -14))) - 10)


out = {'laplacian': [], 'wiener_chain': [], 'smoke': {}}
for size in [(120, 161), (4, 4), (7, 9), (16, 16), (33, 70), (256, 200), (301, 515), (600, 1100), (5, 301), (1000, 9), (64, 2050)]:
    h, w = size
    lum = O.compute_luminance(scene(max(h, 8), max(w, 8), 31))[:h, :w].copy()
    for prm in [(0.25, 1.4, 0.8, 0.2), (0.2, 1.6, 0.7, 0.3), (0.35, 0.5, 1.5, -0.2), (0.2, 1.0, 1.0, 0.0)]:
        got = npy(td.Laplacian(dev, (w, h), td.LaplacianParams(6, *prm)).process(gpu(lum)))
        ref = O.laplacian(lum, *prm)
        d = np.abs(got - ref)
        out['laplacian'].append({'size': size, 'prm': prm, 'max': float(d.max()), 'max_in_half_ulps': float((d / half_ulp(np.maximum(np.abs(got), np.abs(ref)))).max()),
                                 'frac_differing': float((d > 0).mean())})
# 12 MP frame, 11 levels
lum12 = td.compute_luminance(synthetic_rgb(3072, 4096, seed=99, device=dev))
prm12 = (0.2, 1.6, 0.7, 0.3)
got12 = npy(td.Laplacian(dev, (4096, 3072), td.LaplacianParams(6, *prm12)).process(lum12))
ref12 = O.laplacian(npy(lum12), *prm12)
d12 = np.abs(got12 - ref12)
out['laplacian_12mp'] = {'max': float(d12.max()), 'max_in_half_ulps': float((d12 / half_ulp(np.maximum(np.abs(got12), np.abs(ref12)))).max()), 'frac_differing': float((d12 > 0).mean())}
for size in [(97, 131), (192, 256), (64, 64), (300, 420)]:
    h, w = size
    img = scene(h, w, 77)
    ws = td.Wiener(dev, (w, h))
    got = npy(ws.process_log_luminance(gpu(img), 0.075))
    ll = O.compute_luminance(img, True, 1e-4)
    ref = O.modify_luminance(img, O.wiener(ll[:, :, None], 0.075)[:, :, 0], True)
    out['wiener_chain'].append({'size': size, 'max': float(np.abs(got - ref).max())})
# smoke()'s stages
h, w = 192, 256
from torch_darktable.synthetic import synthetic_bayer  # noqa: E402
bayer = synthetic_bayer(h, w, seed=1234, device='cpu')
rgb = td.RCD(dev, (w, h), td.BayerPattern.RGGB).process(bayer.to(dev))
ref_rgb = O.rcd(bayer.numpy(), O.RGGB)
den = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32).process_log_luminance(rgb, 0.075)
ref_den = O.modify_luminance(ref_rgb, O.wiener(O.compute_luminance(ref_rgb, log=True, eps=1e-4)[:, :, None], 0.075, 32, 4)[:, :, 0], log=True)
out['smoke']['wiener'] = float(np.abs(npy(den) - ref_den).max())
loc = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2).process_rgb(den, 0.4)
ref_loc = O.modify_luminance(ref_den, O.bilateral(O.compute_luminance(ref_den), 2.0, 0.2, 0.4))
out['smoke']['bilateral_vs_oracle_chain'] = float(np.abs(npy(loc) - ref_loc).max())
den_np = npy(den)
ref_loc2 = O.modify_luminance(den_np, O.bilateral(O.compute_luminance(den_np), 2.0, 0.2, 0.4))
out['smoke']['bilateral_same_input'] = float(np.abs(npy(loc) - ref_loc2).max())
print(json.dumps(out, indent=1))

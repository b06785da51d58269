"""Measures how far the fp16-storage chain (BASELINE config 3) sits from the fp32 oracle chain on one 12 MP
window, under several error norms -- the data behind the tolerance of
tests/test_gpu_fullsize.py::test_full_pipeline_12mp_fp16_vs_fp32_oracle.  Run on the GPU box."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / 'torch-darktable_amd', ROOT / 'oracle', ROOT):
    sys.path.insert(0, str(p))
import tdk_oracle as O  # noqa: E402
import torch_darktable as td  # noqa: E402
from torch_darktable.synthetic import synthetic_bayer  # noqa: E402

W, H = 4096, 3072
dev = torch.device('cuda', 0)
frame = synthetic_bayer(H, W, seed=1234, device=dev)
out = {}
for y0, x0 in [(1024, 2048), (2000, 304), (64, 3504)]:
    m, n = 64, 192
    res = {}
    for storage in ('f16', 'f32', 'f16 two-stage rgb chain', 'f32 two-stage rgb chain'):
        b = frame.half() if storage.startswith('f16') else frame
        rgb = td.RCD(dev, (W, H), td.BayerPattern.RGGB).process(b)
        if 'two-stage' in storage:  # the intermediate RGB image between denoiser and local contrast materialised (bench.py --chain rgb)
            den = td.Wiener(dev, (W, H)).process_log_luminance(rgb, 0.075)
            loc = td.Bilateral(dev, (W, H), sigma_s=2.0, sigma_r=0.2).process_rgb(den, 0.4)
        else:  # the Lab hand-over (what bench.py and the pipeline run)
            lum, ab = td.Wiener(dev, (W, H)).process_log_luminance_lab(rgb, 0.075)
            loc = td.Bilateral(dev, (W, H), sigma_s=2.0, sigma_r=0.2).process_lab(lum, ab, 0.4, out_dtype=rgb.dtype)
        bw = b[y0 - m:y0 + n + m, x0 - m:x0 + n + m, 0].float().cpu().numpy()
        r = O.rcd(bw, O.RGGB)
        ll = O.compute_luminance(r, True, 1e-4)
        r = O.modify_luminance(r, O.wiener(ll[:, :, None], 0.075, 32, 4)[:, :, 0], True)
        r = O.modify_luminance(r, O.bilateral(O.compute_luminance(r), 2.0, 0.2, 0.4))
        ref = r[m:-m, m:-m]
        got = loc[y0:y0 + n, x0:x0 + n].float().cpu().numpy()
        d = np.abs(got - ref)
        pixmax = np.maximum(ref.max(-1, keepdims=True), 1e-3)
        res[storage] = {
            'max_abs': float(d.max()), 'p999_abs': float(np.quantile(d, 0.999)),
            'max_rel_channel_floor0.05': float((d / np.maximum(np.abs(ref), 0.05)).max()),
            'max_rel_to_pixel_max': float((d / pixmax).max()),
            'max_rel_to_pixel_max_floor0.1': float((d / np.maximum(pixmax, 0.1)).max()),
            'p999_rel_to_pixel_max': float(np.quantile(d / pixmax, 0.999)),
            'ref_min': float(ref.min()), 'ref_max': float(ref.max()),
        }
    out[f'{y0},{x0}'] = res
print(json.dumps(out, indent=1))

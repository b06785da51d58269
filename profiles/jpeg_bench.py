"""Jpeg.encode on the device (csrc/jpeg.hip) at 12 MP: wall time per call (the call synchronises: it returns the host byte stream),
device time per kernel (the library's event timer), stream size, and libjpeg (Pillow) on the host cores beside it.

  python3 profiles/jpeg_bench.py [--size 4096x3072] [--quality 94] [--iters 10]
"""
import argparse
import io
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
import torch_darktable as td  # noqa: E402
from torch_darktable import _native  # noqa: E402
from torch_darktable.synthetic import synthetic_rgb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', default='4096x3072')
    ap.add_argument('--quality', type=int, default=94)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--pillow', action='store_true', help='also time libjpeg (Pillow, optimize=True) on the host')
    a = ap.parse_args()
    w, h = map(int, a.size.split('x'))
    dev = torch.device('cuda', 0)
    rgb = synthetic_rgb(h, w, 5, dev, 0.01)
    u8 = td.aces_tonemap(rgb, td.TonemapParameters(1.0, 0.0, 0.8, 0.0))
    out = {'size': [w, h], 'quality': a.quality, 'git': None}
    for name, sub, prog in (('422', 1, False), ('444', 0, False), ('gray', 2, False), ('422 progressive', 1, True)):
        enc = td.Jpeg()
        data = enc.encode(u8, a.quality, 3, sub, prog)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            data = enc.encode(u8, a.quality, 3, sub, prog)
        wall = (time.perf_counter() - t0) / a.iters
        _native.profile_enable(True, 'tdk_jpeg')
        for _ in range(a.iters):
            enc.encode(u8, a.quality, 3, sub, prog)
        rep = _native.profile_report()
        _native.profile_enable(False)
        kern = {k: round(ms / n * 1e3, 1) for k, (n, ms) in rep.items()}          # us per launch
        per_call = {k: round(ms / a.iters * 1e3, 1) for k, (n, ms) in rep.items()}  # us per encode (a kernel may run once per scan)
        out[name] = {'wall_ms': round(wall * 1e3, 3), 'bytes': int(data.numel()), 'device_us_per_encode': per_call, 'device_us_total': round(sum(per_call.values()), 1),
                     'us_per_launch': kern, 'MP_per_s_wall': round(w * h / wall / 1e6, 1)}
        print(name, json.dumps(out[name]), flush=True)
    if a.pillow:
        from PIL import Image

        src = Image.fromarray(u8.cpu().numpy())
        t0 = time.perf_counter()
        buf = io.BytesIO()
        src.save(buf, 'JPEG', quality=a.quality, optimize=True, subsampling='4:2:2')
        out['pillow_422_host_ms'] = round((time.perf_counter() - t0) * 1e3, 1)
        out['pillow_422_bytes'] = len(buf.getvalue())
        print('pillow', out['pillow_422_host_ms'], 'ms', out['pillow_422_bytes'], 'bytes', flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

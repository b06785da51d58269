"""Generates tests/golden/reference_helpers.npz from the reference's own pure-torch helpers.

Only these functions of the reference run without its CUDA extension:
  torch_darktable/bayer.py:24-47     rgb_to_bayer (+ channels / expand_bayer / stack_bayer)
  torch_darktable/denoise.py:130-158 estimate_channel_noise
  torch_darktable/pipeline/util.py   lerp, normalize_image, resize, resize_longest_edge (the two
                                     @torch.compile decorators are made inert: same eager torch ops)
  torch_darktable/pipeline/transform.py  transform, transformed_size, ImageTransform.next_rotation
They are loaded here file-by-file from /root/reference with inert stand-ins for the modules
they import but do not use on this path (beartype, cv2, the compiled extension).  The outputs
are committed as data; this script only needs re-running if the fixtures are to be regenerated
(the reference tree does not exist on the GPU box).

Everything else in the hot path is a CUDA kernel that cannot be built or run here and the
reference ships no golden vectors for it: those ops stay "parity unpinned" and are anchored on
the literal oracle restatement + source-derived identities (tests/test_oracle_kat.py).
"""

import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path('/root/reference/torch_darktable')
OUT = Path(__file__).resolve().parent / 'reference_helpers.npz'


def _stub_modules():
    bt = types.ModuleType('beartype')
    bt.beartype = lambda f=None, **kw: f if f is not None else (lambda g: g)
    sys.modules['beartype'] = bt
    sys.modules['cv2'] = types.ModuleType('cv2')
    pkg = types.ModuleType('torch_darktable')
    pkg.__path__ = [str(REF)]
    sys.modules['torch_darktable'] = pkg
    ext_mod = types.ModuleType('torch_darktable.extension')

    class _Ext:
        class BayerPattern:
            RGGB, BGGR, GRBG, GBRG = 0x94949494, 0x16161616, 0x61616161, 0x49494949

    ext_mod.extension = _Ext
    sys.modules['torch_darktable.extension'] = ext_mod


def _load(name):
    spec = importlib.util.spec_from_file_location(f'torch_darktable.{name}', REF / f'{name}.py')
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f'torch_darktable.{name}'] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_pipeline(name):
    spec = importlib.util.spec_from_file_location(f'torch_darktable.pipeline.{name}', REF / 'pipeline' / f'{name}.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pipeline_helpers(out, g):
    """Caller-side helpers: fixtures for pipeline/util.py and pipeline/transform.py."""
    real_compile = torch.compile
    torch.compile = lambda f=None, **kw: f if f is not None else (lambda h: h)  # eager: the same torch ops, no inductor build
    try:
        util = _load_pipeline('util')
    finally:
        torch.compile = real_compile
    tr = _load_pipeline('transform')
    img = torch.rand(6, 9, 3, generator=g) * 1.7 - 0.2
    bounds = torch.tensor([-0.15, 1.35])
    out['pipe_img'] = img.numpy()
    out['pipe_bounds'] = bounds.numpy()
    out['pipe_normalized'] = util.normalize_image(img, bounds).numpy()
    out['pipe_lerp'] = util.lerp(img, img.flip(0), 0.3).numpy()
    out['pipe_resized'] = util.resize(img, (4, 5)).numpy()
    sizes = [(4096, 3072), (3072, 4096), (640, 640), (1001, 333)]
    out['pipe_resize_sizes_in'] = np.array(sizes)
    out['pipe_resize_sizes_out'] = np.array([[*util.resize_longest_edge(s, L)] for s in sizes for L in (0, 512, 1000)])
    small = torch.arange(2 * 3 * 2, dtype=torch.float32).view(2, 3, 2)
    out['pipe_transform_in'] = small.numpy()
    for t in tr.ImageTransform:
        out[f'pipe_transform_{t.name}'] = tr.transform(small, t).numpy()
        out[f'pipe_transformed_size_{t.name}'] = np.array(tr.transformed_size((640, 480), t))
        out[f'pipe_next_rotation_{t.name}'] = np.array(t.next_rotation().value)


def main():
    _stub_modules()
    bayer = _load('bayer')
    denoise = _load('denoise')
    out = {}
    g = torch.Generator().manual_seed(20251114)
    for tag, (h, w) in {'small': (8, 8), 'mid': (48, 64)}.items():
        rgb = torch.rand(h, w, 3, generator=g)
        out[f'rgb_{tag}'] = rgb.numpy()
        for pat in bayer.BayerPattern:
            out[f'bayer_{tag}_{pat.name}'] = bayer.rgb_to_bayer(rgb, pat).numpy()
            out[f'pixel_order_{pat.name}'] = np.array(bayer.pixel_order(pat))
            out[f'channels_{pat.name}'] = np.array(bayer.channels(pat))
        out[f'stack_{tag}'] = bayer.stack_bayer(bayer.rgb_to_bayer(rgb)[:, :, 0]).numpy()
    img = torch.rand(64, 64, 3, generator=g) * 0.1 + torch.linspace(0, 1, 64).view(1, 64, 1)
    out['noise_img'] = img.numpy()
    for stride in (1, 8):
        out[f'noise_sigma_stride{stride}'] = denoise.estimate_channel_noise(img, stride).numpy()
    pipeline_helpers(out, g)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, len(out), 'arrays')


if __name__ == '__main__':
    main()

"""Per-op timing table for every SURVEY.md 8(a) op at 12 MP (the bench.py headline is the fused pipeline).

  python profiles/op_bench.py [--storage f16|f32] [--iters 5] > profiles/r01/op_bench_<storage>.json

For each op: device time per call (sum of its kernels, from the library's event timer), the
algorithmic bytes at the op boundary (input + output at the storage type) and the implied GB/s
against the 8 TB/s HBM peak.
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--storage', default='f16')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--height', type=int, default=3072)
    ap.add_argument('--only', default='', help='substring filter on op names (several: a|b)')
    a = ap.parse_args()
    import torch_darktable as td
    from torch_darktable import _native
    from torch_darktable.synthetic import synthetic_bayer

    dev = torch.device('cuda', 0)
    w, h = a.width, a.height
    n = w * h
    dt = torch.float16 if a.storage == 'f16' else torch.float32
    s = 2 if a.storage == 'f16' else 4
    bayer32 = synthetic_bayer(h, w, 1234, dev)
    bayer = bayer32.to(dt)
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    ppg = td.PPG(dev, (w, h), td.BayerPattern.RGGB)
    rgb = rcd.process(bayer).clone()
    rgb32 = rgb.float()
    lum = td.compute_luminance(rgb)
    lum32 = lum.float()
    post = td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, color_smoothing_passes=3, green_eq_local=True, green_eq_global=False)
    wiener = td.Wiener(dev, (w, h), 4, 32)
    wiener16 = td.Wiener(dev, (w, h), 4, 16)
    wiener_ov8 = td.Wiener(dev, (w, h), 8, 32)
    wiener_ov2 = td.Wiener(dev, (w, h), 2, 32)
    bil2 = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    bil8 = td.Bilateral(dev, (w, h), sigma_s=8.0, sigma_r=0.1)
    from torch_darktable.local_contrast import LaplacianParams
    lap = td.Laplacian(dev, (w, h), LaplacianParams(sigma=0.2, shadows=0.8, highlights=1.2, clarity=0.2))
    params = td.TonemapParameters(0.75, 2.0, 1.0, 0.0)
    metrics = td.compute_image_metrics([rgb], 8)
    flat32 = bayer32.reshape(-1).contiguous()
    packed = td.encode12_float(flat32, ids_format=False)
    gains = torch.tensor([1.8, 1.0, 1.5], device=dev)

    ops = {
        # name: (callable, algorithmic bytes per call)
        'bilinear5x5 (f32)': (lambda: td.bilinear5x5_demosaic(bayer32, td.BayerPattern.RGGB), n * (4 + 12)),
        'RCD': (lambda: rcd.process(bayer), n * 4 * s),
        'PPG': (lambda: ppg.process(bayer), n * 4 * s),
        'PostProcess(3 smoothing + local eq) (f32)': (lambda: post.process(rgb32), n * 24),
        'apply_white_balance (f32)': (lambda: td.apply_white_balance(bayer32.squeeze(-1), gains, td.BayerPattern.RGGB), n * 8),
        'compute_luminance': (lambda: td.compute_luminance(rgb), n * 4 * s),
        'modify_luminance': (lambda: td.modify_luminance(rgb, lum), n * 7 * s),
        'Wiener.process C=1 (K=32, ov=4)': (lambda: wiener.process(lum.unsqueeze(2), 0.075), n * 2 * s),
        'Wiener.process C=1 (K=16, ov=4)': (lambda: wiener16.process(lum.unsqueeze(2), 0.075), n * 2 * s),
        'Wiener.process C=1 (K=32, ov=8)': (lambda: wiener_ov8.process(lum.unsqueeze(2), 0.075), n * 2 * s),
        'Wiener.process C=1 (K=32, ov=2)': (lambda: wiener_ov2.process(lum.unsqueeze(2), 0.075), n * 2 * s),
        'Wiener.process C=3 (K=32, ov=4)': (lambda: wiener.process(rgb, 0.05), n * 6 * s),
        'Wiener.process_log_luminance': (lambda: wiener.process_log_luminance(rgb, 0.075), n * 6 * s),
        'Bilateral.process (2.0, 0.2)': (lambda: bil2.process(lum, 0.4), n * 2 * s),
        'Bilateral.process (8.0, 0.1)': (lambda: bil8.process(lum, 0.4), n * 2 * s),
        'Bilateral.process_rgb (2.0, 0.2)': (lambda: bil2.process_rgb(rgb, 0.4), n * 6 * s),
        'Laplacian.process (f32)': (lambda: lap.process(lum32), n * 8),
        'compute_image_metrics (stride 8)': (lambda: td.compute_image_metrics([rgb], 8), n // 64 * 3 * s),
        'reinhard_tonemap': (lambda: td.reinhard_tonemap(rgb, metrics, params), n * (3 * s + 3)),
        'aces_tonemap': (lambda: td.aces_tonemap(rgb, params), n * (3 * s + 3)),
        'rgb_to_lab (f32)': (lambda: td.rgb_to_lab(rgb32), n * 24),
    }
    if a.storage == 'f16':  # the ops that accept binary16 storage since round 5 (fp32 arithmetic, one rounding at the store)
        ops['PostProcess(3 smoothing + local eq) (f16 storage)'] = (lambda: post.process(rgb), n * 12)
        ops['apply_white_balance (f16 storage)'] = (lambda: td.apply_white_balance(bayer.squeeze(-1), gains, td.BayerPattern.RGGB), n * 4)
        ops['Laplacian.process (f16 storage)'] = (lambda: lap.process(lum), n * 4)
        ops['rgb_to_lab (f16 storage)'] = (lambda: td.rgb_to_lab(rgb), n * 12)
    u8 = td.reinhard_tonemap(rgb, metrics, params)
    coder = td.Jpeg()
    ops['Jpeg.encode (4:2:2, q94; device kernels, the call also synchronises and copies the stream to the host)'] = (lambda: coder.encode(u8), n * 3)
    ops['decode12 -> f32'] = (lambda: td.decode12_float(packed, ids_format=False), n * 1.5 + n * 4)
    ops['decode12 -> f16'] = (lambda: td.decode12_half(packed, ids_format=False), n * 1.5 + n * 2)
    ops['encode12 <- f32'] = (lambda: td.encode12_float(flat32, ids_format=False), n * 1.5 + n * 4)

    # the caller: packed 12-bit raw bytes -> ImageProcessor -> uint8
    try:
        from torch_darktable.pipeline import CameraSettings, ImageProcessingSettings, ImageProcessor, ImageTransform, ToneMapper
        settings = ImageProcessingSettings(tone_gamma=0.75, tone_intensity=2.0, moving_average=1.0, enable_bilateral=True, tone_mapping=ToneMapper.reinhard)
        cam = CameraSettings(name='cam', image_size=(w, h), padding=0, white_balance=(1.5, 1.0, 1.2), image_processing=settings, transform=ImageTransform.none)
        proc = ImageProcessor.from_camera_settings(cam, dev)
        raw = packed.clone()
        ops['ImageProcessor.process (packed12 -> u8, defaults + bilateral)'] = (lambda: proc.process(raw, 'cam'), n * 1.5 + n * 3)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({'op': 'ImageProcessor.process', 'error': str(e)[:200]}), flush=True)

    rows = []
    for name, (fn, nbytes) in ops.items():
        if a.only and not any(t.strip().lower() in name.lower() for t in a.only.split('|')):
            continue
        try:
            fn()
            torch.cuda.synchronize()
            _native.profile_enable(True)
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
            rep = _native.profile_report()
            _native.profile_enable(False)
            ms = sum(v[1] for v in rep.values()) / a.iters
            rows.append({'op': name, 'ms': round(ms, 4), 'kernels_per_call': sum(v[0] for v in rep.values()) // a.iters,
                         'algorithmic_MB': round(nbytes / 1e6, 1), 'GBps': round(nbytes / ms / 1e6, 1), 'frac_of_8TBps': round(nbytes / ms / 1e6 / 8000, 4),
                         'MPps': round(n / ms / 1e3, 0), 'kernels': {k: round(v[1] / a.iters, 4) for k, v in rep.items()}})
        except Exception as e:  # noqa: BLE001
            rows.append({'op': name, 'error': str(e)[:200]})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == '__main__':
    main()

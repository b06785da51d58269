"""Run one op of the hot path a few times on synthetic 12 MP data -- the target of the
rocprofv3 passes whose summaries are committed in this directory.

  python profiles/run_op.py {rcd,ppg,postprocess,wiener,bilateral,laplacian,tonemap,luminance,jpeg,isp,isp_jpeg} [--iters N] [--storage f16|f32]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('op')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--storage', default='f16')
    ap.add_argument('--width', type=int, default=4096)
    ap.add_argument('--height', type=int, default=3072)
    a = ap.parse_args()
    import torch_darktable as td
    from torch_darktable.synthetic import synthetic_bayer

    dev = torch.device('cuda', 0)
    w, h = a.width, a.height
    dt = torch.float16 if a.storage == 'f16' else torch.float32
    bayer = synthetic_bayer(h, w, 1234, dev).to(dt)
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    rgb = rcd.process(bayer)
    ppg = td.PPG(dev, (w, h), td.BayerPattern.RGGB)
    rgb32 = rgb.float()
    post = td.PostProcess(dev, (w, h), td.BayerPattern.RGGB, color_smoothing_passes=3, green_eq_local=True, green_eq_global=False)
    lum = td.compute_luminance(rgb)
    loglum = td.compute_log_luminance(rgb, 1e-4)
    wiener = td.Wiener(dev, (w, h), 4, 32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    lap = td.Laplacian(dev, (w, h), td.LaplacianParams(6, 0.2, 1.6, 0.7, 0.3)) if a.op == 'laplacian' else None
    lum32 = lum.float()
    params = td.TonemapParameters(0.75, 2.0, 1.0, 0.0)
    metrics = td.compute_image_metrics([rgb], 8)
    lum_plane = torch.empty((h, w), dtype=torch.float32, device=dev)
    ab_plane = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    jpeg = td.Jpeg()
    u8 = td.reinhard_tonemap(rgb, metrics, params)
    torch.cuda.synchronize()
    for _ in range(a.iters):
        if a.op == 'rcd':
            rcd.process(bayer)
        elif a.op == 'ppg':
            ppg.process(bayer)
        elif a.op == 'postprocess':
            post.process(rgb32)
        elif a.op == 'wiener':
            wiener.process(loglum.unsqueeze(2), 0.075)
        elif a.op == 'bilateral':
            bil.process(lum, 0.4)
        elif a.op == 'laplacian':
            lap.process(lum32)
        elif a.op == 'tonemap':
            td.reinhard_tonemap(rgb, metrics, params)
        elif a.op == 'luminance':
            td.modify_luminance(rgb, td.compute_luminance(rgb))
        elif a.op == 'isp':  # the chain bench.py times (with its stage hand-overs; the kernels it runs with frames on several streams)
            with td.torch_darktable_extension.concurrent_frames():
                x = rcd.process(bayer)
            wiener.process_log_luminance_lab(x, 0.075, luminance_out=lum_plane, chroma_out=ab_plane)
            x = bil.process_lab(lum_plane, ab_plane, 0.4, out_dtype=x.dtype, metrics=acc)
            td.reinhard_tonemap(x, acc.finish(), params)
        elif a.op == 'isp_jpeg':  # the chain + the format after it: Jpeg.encode of the tone-mapped frame (4:2:2, quality 94, baseline)
            with td.torch_darktable_extension.concurrent_frames():
                x = rcd.process(bayer)
            wiener.process_log_luminance_lab(x, 0.075, luminance_out=lum_plane, chroma_out=ab_plane)
            x = bil.process_lab(lum_plane, ab_plane, 0.4, out_dtype=x.dtype, metrics=acc)
            jpeg.encode(td.reinhard_tonemap(x, acc.finish(), params))
        elif a.op == 'jpeg':
            jpeg.encode(u8)
        elif a.op == 'isp_rgb':  # the same chain with the intermediate RGB image materialised (bench.py --chain rgb)
            with td.torch_darktable_extension.concurrent_frames():
                x = rcd.process(bayer)
            x = wiener.process_log_luminance(x, 0.075, luminance_out=lum_plane)
            x = bil.process_rgb(x, 0.4, luminance=lum_plane, metrics=acc)
            td.reinhard_tonemap(x, acc.finish(), params)
        else:
            raise SystemExit(f'unknown op {a.op}')
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()

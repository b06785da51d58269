"""Host-buffer hand-over rate: 12 MP packed-12 raw FILES -> ImageProcessor -> uint8 on the device,
(a) the reference's way (load_raw_bytes per frame on the caller's thread) and (b) RawFrameStream
(reader thread, pinned ring, copy stream).  Never part of bench.py's `value` (inputs there are
device-resident); quoted in DESIGN.md section 5.

  python profiles/stream_bench.py [--frames 32] [--depth 3]
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--depth', type=int, default=3)
    a = ap.parse_args()
    import torch_darktable as td
    from torch_darktable.pipeline import CameraSettings, ImageProcessingSettings, ImageProcessor, ImageTransform, RawFrameStream, ToneMapper
    from torch_darktable.pipeline.camera_settings import load_raw_bytes
    from torch_darktable.synthetic import synthetic_bayer

    dev = torch.device('cuda', 0)
    w, h = 4096, 3072
    settings = ImageProcessingSettings(tone_gamma=0.75, tone_intensity=2.0, moving_average=1.0, enable_bilateral=True, tone_mapping=ToneMapper.reinhard)
    cam = CameraSettings(name='cam', image_size=(w, h), padding=0, white_balance=(1.5, 1.0, 1.2), image_processing=settings, transform=ImageTransform.none)
    proc = ImageProcessor.from_camera_settings(cam, dev)
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i in range(4):  # four distinct frames, cycled (page cache warm: this measures the upload path, not the disk)
            packed = td.encode12_float(synthetic_bayer(h, w, 1234 + i, dev).reshape(-1).contiguous(), ids_format=False)
            p = Path(d) / f'f{i}.raw'
            p.write_bytes(packed.cpu().numpy().tobytes())
            paths.append(p)
        files = [paths[i % 4] for i in range(a.frames)]
        nbytes = cam.bytes
        out = {}
        variants = {'load_raw_bytes (serial)': None}
        for r in (1, 2, 4):
            variants[f'RawFrameStream depth={a.depth} readers={r}'] = r
        for name, readers in variants.items():
            for rep in range(2):  # first repetition = warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if name.startswith('load'):
                    for f in files:
                        proc.process(load_raw_bytes(f, dev), 'cam')
                else:
                    for frame in RawFrameStream(files, dev, nbytes, depth=a.depth, readers=readers):
                        proc.process(frame, 'cam')
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out[name] = {'ms_per_frame': round(dt / a.frames * 1e3, 3), 'MPps': round(a.frames * w * h / 1e6 / dt, 1)}
        print(json.dumps({'frames': a.frames, 'frame_bytes': nbytes, 'pipeline': 'ImageProcessor fp32 (decode, WB, RCD, Wiener, bilateral, Reinhard)', **out}))


if __name__ == '__main__':
    main()

"""Timing experiment (wrong results): what the 12 MP chain would gain if the RCD strips emitted log-lightness + Lab chroma themselves
(no lum_lab_extract launch, no RGB image between demosaic and denoiser).
    python profiles/build_experiments.py -DTDK_RQ_FAKE_LAB      # variants/exp.so
    TDK_LIB_PATH=variants/exp.so python profiles/rcd_lab_fusion_exp.py
Runs the bench's chain on three streams twice: as it is, and with (a) an RCD launch whose store phase converts its four pixels to
(log L, a, b) and writes 12 bytes per pixel and (b) the extraction launch skipped (TDK_FAKE_SKIP_EXTRACT)."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'torch-darktable_amd'))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import torch_darktable as td  # noqa: E402
from torch_darktable import _native  # noqa: E402
from torch_darktable.sharding import FrameStreams  # noqa: E402
from torch_darktable.synthetic import synthetic_bayer  # noqa: E402

dev = torch.device('cuda', 0)
w, h, frames = 4096, 3072, 8
inputs = [synthetic_bayer(h, w, seed=1234 + i, device=dev).half() for i in range(frames)]


def make_chain(fused):
    rcd = td.RCD(dev, (w, h), td.BayerPattern.RGGB)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    params = td.TonemapParameters(0.75, 2.0, 1.0, 0.0)
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    ab = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    fat = torch.empty((h, w, 3), dtype=torch.float32, device=dev)  # 12 bytes per pixel for the fake Lab output
    dummy = torch.empty((h, w, 3), dtype=torch.float16, device=dev)  # never read: the extraction launch is skipped

    def frame(bayer):
        if fused:
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _native.check(_native.lib.tdk_rcd_ex(C.c_void_p(bayer.data_ptr()), C.c_void_p(fat.data_ptr()), None, w, h, 0x94949494, 1, 2, s))
            x = dummy
        else:
            x = rcd.process(bayer)
        wiener.process_log_luminance_lab(x, 0.075, luminance_out=lum, chroma_out=ab)
        out = bil.process_lab(lum, ab, 0.4, out_dtype=torch.float16, metrics=acc)
        return td.reinhard_tonemap(out, acc.finish(), params)

    return frame


def run(fused, label):
    if fused:
        os.environ['TDK_FAKE_SKIP_EXTRACT'] = '1'
    else:
        os.environ.pop('TDK_FAKE_SKIP_EXTRACT', None)
    runner = FrameStreams(dev, lambda: make_chain(fused), streams=3)
    for _ in range(5):
        runner.issue(inputs)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            runner.issue(inputs)
        torch.cuda.synchronize()
        res.append(frames * 20 * w * h / 1e6 / (time.perf_counter() - t0))
    print(label, [round(r) for r in res], 'MP/s', flush=True)


# The two arms need different LIBRARIES: a -DTDK_RQ_FAKE_LAB build writes 12 bytes per pixel from EVERY float16 RCD launch, so the plain chain
# (6 bytes per pixel buffers) must never run on it.
#   python profiles/rcd_lab_fusion_exp.py plain                                   (in-tree library)
#   TDK_LIB_PATH=variants/exp.so python profiles/rcd_lab_fusion_exp.py fused      (variants/exp.so built with -DTDK_RQ_FAKE_LAB)
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
if mode == 'fused':
    assert os.environ.get('TDK_LIB_PATH'), 'the fused arm needs the -DTDK_RQ_FAKE_LAB library'
    run(True, 'RCD emits Lab (timing only)')
else:
    assert not os.environ.get('TDK_LIB_PATH'), 'the plain arm must run on the in-tree library'
    run(False, 'chain as it is            ')

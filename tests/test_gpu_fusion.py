"""GPU: the stage hand-overs of the pipeline (lightness plane from the denoiser to the bilateral stage, image
metrics fed through a MetricsAccumulator) give the SAME results as the separate wrapper calls they replace
(reference pipeline/image_processor.py:257-300 runs the separate calls)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('size', [(192, 256), (250, 334), (1024, 1536)])
def test_luminance_handover_and_metrics_accumulator_equal_the_chain(td, dev, dtype, size):
    from torch_darktable.synthetic import synthetic_rgb

    h, w = size
    rgb = synthetic_rgb(h, w, seed=61, device=dev).to(dtype)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)

    # the chain of separate calls
    den = wiener.process_log_luminance(rgb, 0.075)
    loc = bil.process_rgb(den, 0.4)
    metrics = td.compute_image_metrics([loc], stride=8)

    # with hand-overs
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    den2 = wiener.process_log_luminance(rgb, 0.075, luminance_out=lum)
    assert torch.equal(den2, den)
    assert torch.equal(lum, td.extension.extension._extract_luminance(den, False, 1e-6, torch.float32))   # == compute_luminance(den) as fp32
    loc2 = bil.process_rgb(den2, 0.4, luminance=lum, metrics=acc)
    assert torch.equal(loc2, loc)
    m2 = acc.finish()
    assert torch.allclose(m2, metrics, rtol=2e-5, atol=1e-7)      # float atomics: order of the sums differs
    assert torch.count_nonzero(acc.acc).item() == 0               # the accumulator is clean for the next frame
    # a second frame through the same accumulator
    loc3 = bil.process_rgb(den2, 0.4, luminance=lum, metrics=acc)
    assert torch.equal(loc3, loc) and torch.allclose(acc.finish(), metrics, rtol=2e-5, atol=1e-7)


def test_metrics_accumulator_equals_compute_image_metrics(td, dev):
    from torch_darktable.synthetic import synthetic_rgb

    imgs = [synthetic_rgb(96, 131, seed=70 + i, device=dev) * 1.2 for i in range(3)]
    acc = td.tonemap.MetricsAccumulator(dev, stride=4, min_gray=1e-3)
    for im in imgs:
        acc.add(im)
    assert torch.allclose(acc.finish(), td.compute_image_metrics(imgs, stride=4, min_gray=1e-3), rtol=2e-5, atol=1e-7)


def test_metrics_accumulator_on_the_grid_path(td, dev):
    """sigma_s = 8 uses the four-kernel grid path; the accumulator is fed the same way."""
    from torch_darktable.synthetic import synthetic_rgb

    rgb = synthetic_rgb(160, 224, seed=5, device=dev)
    bil = td.Bilateral(dev, (224, 160), sigma_s=8.0, sigma_r=0.1)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    out = bil.process_rgb(rgb, 0.3, metrics=acc)
    assert torch.equal(out, bil.process_rgb(rgb, 0.3))
    assert torch.allclose(acc.finish(), td.compute_image_metrics([out], stride=8), rtol=2e-5, atol=1e-7)


def test_workspace_scratch_is_per_stream(td, dev):
    """One workspace object used from two streams: each stream gets its own scratch (slabs / planes), results agree."""
    from torch_darktable.synthetic import synthetic_rgb

    rgb = synthetic_rgb(256, 320, seed=9, device=dev)
    wiener = td.Wiener(dev, (320, 256))
    ref = wiener.process_log_luminance(rgb, 0.05)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    for s in (s1, s2, s1, s2):
        with torch.cuda.stream(s):
            outs.append(wiener.process_log_luminance(rgb, 0.05))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert len(wiener._wiener._scratch) == 3


def test_batch_on_two_streams_equals_back_to_back(td, dev):
    """sharding.FrameStreams (used by bench.py) issues the frames of a batch round-robin on two HIP streams, each with
    its own chain object (op workspaces, accumulators), so that kernels of neighbouring frames overlap: every frame must
    come out bit-identical to the same chain run back to back on one stream."""
    import bench
    from torch_darktable.sharding import FrameStreams
    from torch_darktable.synthetic import synthetic_bayer

    w, h, frames = 1536, 1024, 6
    inputs = [synthetic_bayer(h, w, seed=900 + i, device=dev).to(torch.float16) for i in range(frames)]
    make = lambda: bench.build_pipeline(td, dev, w, h, 'f16', 'isp')[1]
    serial = make()
    ref = [serial(b).clone() for b in inputs]
    runner = FrameStreams(dev, make, streams=2)
    for _ in range(3):   # repeated: a race would not hit every time
        outs = runner.run(inputs)          # joined: usable on the current stream
        for i in range(frames):
            assert torch.equal(outs[i], ref[i]), f'frame {i} differs between the two-stream and the one-stream run'
    one = FrameStreams(dev, make, streams=1).run(inputs)
    assert all(torch.equal(a, b) for a, b in zip(one, ref))

"""GPU parity of the device JPEG encoder (csrc/jpeg.hip, SURVEY.md section 8 f-4: reference csrc/jpeg_encoder.cu:104-180) through the
drop-in surface `torch_darktable.Jpeg.encode`: the byte stream is IDENTICAL to the CPU restatement's (oracle/src/jpeg.c), which
tests/test_oracle_jpeg.py pins against libjpeg; here libjpeg also decodes the device streams themselves.  Parity with nvjpeg's own
bytes is unpinned (closed library, no JPEG output in the reference)."""

import io

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a visible MI355X'
    return torch.device('cuda', 0)


def sample_image(h, w, seed=0, noise=6.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 17.0) * np.cos(yy / 23.0), 127 + 90 * np.sin((xx + yy) / 31.0), 127 + 80 * np.cos(xx / 11.0 - yy / 7.0)], -1)
    return np.clip(img + rng.normal(0, noise, img.shape), 0, 255).astype(np.uint8)


def arrange(img, fmt):
    """RGB (H, W, 3) -> the layout of input format fmt (0 BGR planar, 1 RGB planar, 2 BGRI, 3 RGBI)."""
    a = img if fmt & 1 else img[:, :, ::-1]
    return np.ascontiguousarray(a.transpose(2, 0, 1) if fmt < 2 else a)


def decode(stream):
    im = Image.open(io.BytesIO(bytes(stream)))
    im.load()
    return im


@pytest.mark.parametrize('h,w', [(1, 1), (3, 2), (8, 8), (17, 9), (64, 96), (203, 331), (100, 2100), (520, 1030)])
@pytest.mark.parametrize('sub', [0, 1, 2])
@pytest.mark.parametrize('progressive', [False, True])
def test_stream_identical_to_oracle(td, oracle, dev, h, w, sub, progressive):
    img = sample_image(h, w, h * 7 + w)
    enc = td.Jpeg()
    for quality, fmt in ((94, 3), (35, 2), (100, 1), (75, 0)):
        x = torch.from_numpy(arrange(img, fmt)).to(dev)
        got = enc.encode(x, quality, fmt, sub, progressive).numpy()
        want = oracle.jpeg_encode(arrange(img, fmt), quality, fmt, sub, progressive)
        assert got.shape == want.shape and np.array_equal(got, want), (quality, fmt, got.shape, want.shape)
    im = decode(got)
    assert im.size == (w, h) and im.mode == ('L' if sub == 2 else 'RGB')


@pytest.mark.parametrize('sub', [0, 1, 2])
def test_coefficients_identical_to_oracle(td, oracle, dev, sub):
    """The first pass alone (colour conversion, FDCT, quantisation): the same int16 coefficients block for block."""
    from torch_darktable._native import lib
    from torch_darktable.torch_darktable_extension import _ptr, _stream, check

    h, w = 120, 1100   # more than one strip tile across for every subsampling except gray
    img = sample_image(h, w, 3, noise=25.0)
    enc = td.Jpeg()
    enc.encode(torch.from_numpy(img).to(dev), 97, 3, sub, False)
    _, want = oracle.jpeg_encode(img, 97, 3, sub, False, return_coefs=True)
    got = torch.empty(want.size, dtype=torch.int16)
    check(lib.tdk_jpeg_coefficients(_ptr(enc.jpeg._workspace[1]), w, h, sub, _ptr(got), _stream()))
    assert np.array_equal(got.numpy(), want)


def test_extreme_images_identical_and_valid(td, oracle, dev):
    rng = np.random.default_rng(11)
    enc = td.Jpeg()
    for img in (np.zeros((17, 9, 3), np.uint8), np.full((8, 8, 3), 255, np.uint8), rng.integers(0, 256, (40, 56, 3), dtype=np.uint8),
                (rng.integers(0, 2, (333, 470, 3)) * 255).astype(np.uint8)):
        for quality in (1, 100):
            for sub in (0, 1, 2):
                for progressive in (False, True):
                    got = enc.encode(torch.from_numpy(img).to(dev), quality, 3, sub, progressive).numpy()
                    assert np.array_equal(got, oracle.jpeg_encode(img, quality, 3, sub, progressive)), (img.shape, quality, sub, progressive)
                    assert decode(got).size == (img.shape[1], img.shape[0])


def test_api_errors_and_result_type(td, dev):
    enc = td.Jpeg()
    img = torch.from_numpy(sample_image(32, 48)).to(dev)
    data = enc.encode(img, quality=92, input_format=td.InputFormat.RGBI, subsampling=td.Subsampling.CSS_422, progressive=False)
    assert data.device.type == 'cpu' and data.dtype == torch.uint8 and data.dim() == 1 and bytes(data[:2].tolist()) == b'\xff\xd8'
    assert torch.equal(td.Jpeg().encode(img), enc.encode(img, 94, 3, 1, False))   # the wrapper's defaults (jpeg.py:25-28)
    for bad, match in ((img.cpu(), 'CUDA'), (img.float(), 'uint8'), (img.permute(1, 0, 2), 'contiguous'), (img[:, :, :2].contiguous(), 'interleaved'),
                       (img, 'planar')):
        with pytest.raises(RuntimeError, match=match):
            enc.encode(bad, 90, td.InputFormat.RGB if match == 'planar' else td.InputFormat.RGBI, td.Subsampling.CSS_444, False)
    with pytest.raises(RuntimeError):
        enc.encode(img, 90, 7, td.Subsampling.CSS_444, False)
    # not on the current stream's device guard: a second coder object, another size, reuses nothing of the first
    other = td.Jpeg().encode(torch.from_numpy(sample_image(40, 40)).to(dev), 80, 3, 0, True)
    assert decode(other.numpy()).size == (40, 40)


def test_tonemap_output_feeds_the_encoder_12mp(td, oracle, dev):
    """The hot path's last stage hands its uint8 (H, W, 3) result to the encoder on the device; whole 12 MP frame, the stream
    identical to the oracle's, decoded by libjpeg, as close to the source as libjpeg's own encoding."""
    from torch_darktable.synthetic import synthetic_rgb

    h, w = 3072, 4096
    rgb = synthetic_rgb(h, w, 5, dev, 0.01)
    u8 = td.aces_tonemap(rgb, td.TonemapParameters(1.0, 0.0, 0.8, 0.0))
    data = td.Jpeg().encode(u8, 94, td.InputFormat.RGBI, td.Subsampling.CSS_422, False).numpy()
    src = u8.cpu().numpy()
    assert np.array_equal(data, oracle.jpeg_encode(src, 94, 3, 1, False))
    dec = np.asarray(decode(data)).astype(np.float64)
    buf = io.BytesIO()
    Image.fromarray(src).save(buf, 'JPEG', quality=94, optimize=True, subsampling='4:2:2')
    ref = np.asarray(Image.open(io.BytesIO(buf.getvalue()))).astype(np.float64)
    mse_ours, mse_ref = np.mean((dec - src) ** 2), np.mean((ref - src) ** 2)
    assert mse_ours <= mse_ref * 1.02 and len(data) <= len(buf.getvalue()) * 1.01, (mse_ours, mse_ref, len(data), len(buf.getvalue()))


def test_side_stream_and_50mp(td, oracle, dev):
    """The coder works on PyTorch's current stream (the reference: at::cuda::getCurrentCUDAStream, jpeg_encoder.cu:156) -- here a side
    stream whose producer is still running -- and at BASELINE config 5's frame size (8192 x 6144: 1.57 M blocks, 0.87 GB of workspace)."""
    from torch_darktable.synthetic import synthetic_rgb

    h, w = 6144, 8192
    side = torch.cuda.Stream(device=dev)
    rgb = synthetic_rgb(h, w, 9, dev, 0.01)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        u8 = td.aces_tonemap(rgb, td.TonemapParameters(1.0, 0.0, 0.8, 0.0))   # producer on the same side stream, not synchronised
        data = td.Jpeg().encode(u8, 90, td.InputFormat.RGBI, td.Subsampling.CSS_422, False).numpy()
    side.synchronize()
    want = oracle.jpeg_encode(u8.cpu().numpy(), 90, 3, 1, False)
    assert data.shape == want.shape and np.array_equal(data, want)
    assert decode(data).size == (w, h)


def test_concurrent_coders_on_their_own_streams(td, oracle, dev):
    """Three host threads, each with its own coder object and HIP stream (how profiles/raw_to_jpeg.py drives a GPU): the library keeps
    no process-global state, every call synchronises only its own stream, and each stream still equals the oracle's bytes."""
    import threading

    imgs = [sample_image(200 + 8 * k, 300 + 16 * k, 40 + k) for k in range(3)]
    want = [oracle.jpeg_encode(im, 90, 3, k % 3, k == 1) for k, im in enumerate(imgs)]
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream(device=dev)
            coder = td.Jpeg()
            x = torch.from_numpy(imgs[k]).to(dev)
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(stream):
                for _ in range(8):
                    got = coder.encode(x, 90, td.InputFormat.RGBI, k % 3, k == 1).numpy()
                    if not np.array_equal(got, want[k]):
                        errors.append((k, got.shape, want[k].shape))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors

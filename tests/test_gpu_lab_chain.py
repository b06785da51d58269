"""GPU: the Lab hand-over chain (include/tdk_hip.h: tdk_compute_log_luminance_lab, tdk_wiener_log_luminance_lab, tdk_bilateral_lab).

Wiener.process_log_luminance -> Bilateral.process_rgb of the reference are two Lab round trips of the same pixel
(torch_darktable/denoise.py:54-58, local_contrast.py:109-114); the hand-over carries the pixel as (L plane, (a, b) plane) between
the stages and converts back once.  Parity: against the CPU oracle's two-stage chain and against this library's own two-stage
chain, within the tolerance of the colour operators (2e-5 absolute: hardware exp2 / log2 against libm, one sRGB encode / decode
round trip skipped) -- including pixels the reference's clip to [0, 1] changes, whose (L, a, b) the finish kernel re-derives."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


def npy(t):
    return t.detach().float().cpu().numpy()


def saturated_scene(scene, h, w, seed):
    """The synthetic scene with blocks of saturated / out-of-gamut-prone colours and a block above 1: pixels whose lightness
    replacement leaves the sRGB gamut (the clipped path of both stages)."""
    rgb = scene(h, w, seed).copy()
    rgb[10:40, 10:60] = [0.95, 0.02, 0.03]
    rgb[50:80, 20:90] = [0.02, 0.03, 0.97]
    rgb[90:120, 5:70] = [0.01, 0.9, 0.02]
    rgb[30:60, 100:140] = [1.3, 0.4, 0.2]     # a channel above 1: compute_log_luminance clips it, modify_log_luminance does not
    rgb[70:100, 100:150] = [0.0, 0.0, 0.0]
    rng = np.random.default_rng(seed)
    rgb[:128, :160] += rng.normal(0, 0.01, (128, 160, 3)).astype(np.float32) * (rgb[:128, :160] > 0)
    return np.maximum(rgb, 0).astype(np.float32)


def oracle_chain(oracle, rgb, sigma_s, sigma_r):
    ll = oracle.compute_luminance(rgb, True, 1e-4)
    den = oracle.modify_luminance(rgb, oracle.wiener(ll[:, :, None], 0.075, 32, 4)[:, :, 0], True)
    lum = oracle.compute_luminance(den)
    return den, lum, oracle.modify_luminance(den, oracle.bilateral(lum, sigma_s, sigma_r, 0.4))


def test_log_luminance_lab_extract(td, oracle, dev, scene):
    from torch_darktable import _native
    from torch_darktable.torch_darktable_extension import _ptr, _stream

    h, w = 128, 160
    rgb = saturated_scene(scene, h, w, 3)
    for dt in (torch.float32, torch.float16):
        x = torch.from_numpy(rgb).to(dev).to(dt)
        ll = torch.empty((h, w), dtype=torch.float32, device=dev)
        ab = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
        _native.check(_native.lib.tdk_compute_log_luminance_lab(_ptr(x), _ptr(ll), _ptr(ab), h * w, 1e-4, None, 0 if dt == torch.float32 else 1, _stream()))
        xr = npy(x)
        assert np.abs(npy(ll) - oracle.compute_luminance(xr, True, 1e-4)).max() <= TOL
        lab = oracle.color_op('rgb_to_lab', xr)
        assert np.abs(npy(ab) - lab[:, :, 1:]).max() <= TOL
        # == the library's own operators
        assert torch.allclose(ll, td.compute_log_luminance(x.float(), 1e-4), atol=2e-6, rtol=0)


@pytest.mark.parametrize('sigmas', [(2.0, 0.2), (8.0, 0.1)])   # LDS tile kernel / four-kernel grid path
@pytest.mark.parametrize('size', [(128, 160), (250, 334)])
def test_lab_chain_against_the_oracle_chain(td, oracle, dev, scene, size, sigmas):
    h, w = size
    rgb = saturated_scene(scene, 128, 160, 5) if size == (128, 160) else scene(h, w, 6)
    x = torch.from_numpy(rgb).to(dev)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=sigmas[0], sigma_r=sigmas[1])
    lum, ab = wiener.process_log_luminance_lab(x, 0.075)
    out = bil.process_lab(lum, ab, 0.4)
    den_ref, lum_ref, out_ref = oracle_chain(oracle, rgb, *sigmas)
    assert lum.shape == (h, w) and ab.shape == (h, w, 2) and out.shape == (h, w, 3) and out.dtype == torch.float32
    assert np.abs(npy(lum) - lum_ref).max() <= TOL, np.abs(npy(lum) - lum_ref).max()
    assert np.abs(npy(ab) - oracle.color_op('rgb_to_lab', den_ref)[:, :, 1:]).max() <= 2 * TOL
    d = np.abs(npy(out) - out_ref)
    assert d.max() <= 2 * TOL, (d.max(), np.unravel_index(d.argmax(), d.shape))
    if size == (128, 160):  # the clipped path was taken: some pixels of the intermediate image sit on the gamut boundary
        assert ((den_ref == 0) | (den_ref == 1)).any(-1).mean() > 0.01
    # and the library's two-stage chain
    two = bil.process_rgb(wiener.process_log_luminance(x, 0.075), 0.4)
    assert (out - two).abs().max().item() <= 2 * TOL


@pytest.mark.parametrize('size', [(192, 256), (1024, 1536)])
def test_lab_chain_float16_storage(td, dev, size):
    """float16 images in and out: the hand-over planes stay float32, so the chain rounds to binary16 once less than the
    two-stage chain; both must agree within the binary16 rounding of the intermediate image."""
    from torch_darktable.synthetic import synthetic_rgb

    h, w = size
    rgb = synthetic_rgb(h, w, seed=62, device=dev)
    x16 = rgb.half()
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    acc = td.tonemap.MetricsAccumulator(dev, stride=8)
    lum = torch.empty((h, w), dtype=torch.float32, device=dev)
    ab = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    l2, a2 = wiener.process_log_luminance_lab(x16, 0.075, luminance_out=lum, chroma_out=ab)
    assert l2 is lum and a2 is ab
    out16 = bil.process_lab(lum, ab, 0.4, out_dtype=torch.float16, metrics=acc)
    assert out16.dtype == torch.float16
    # the float32 chain on the same (binary16-valued) input: the only difference is the final rounding
    ref32 = bil.process_lab(*wiener.process_log_luminance_lab(x16.float(), 0.075), 0.4)
    assert torch.equal(out16, ref32.half())
    two = bil.process_rgb(wiener.process_log_luminance(x16, 0.075), 0.4)
    rel = ((out16.float() - two.float()).abs() / two.float().amax(-1, keepdim=True).clamp_min(0.05)).max().item()
    assert rel <= 1.5e-3, rel
    m = acc.finish()
    assert torch.allclose(m, td.compute_image_metrics([out16], stride=8), rtol=2e-5, atol=1e-7)


def test_lab_chain_with_the_pipelines_normalisation_folded_in(td, dev):
    """bounds=: normalize_image(image, bounds) (reference pipeline/util.py:8-10) applied while the first kernel of the chain reads the
    image.  float32 images: the very bits of normalising first (same IEEE expression); float16 images: the chain skips the binary16
    rounding of the normalised image, so it agrees with the two-step form within that rounding."""
    from torch_darktable.pipeline.util import normalize_image
    from torch_darktable.synthetic import synthetic_rgb

    h, w = 192, 256
    rgb = synthetic_rgb(h, w, seed=63, device=dev) * 1.7 + 0.05
    bounds = torch.tensor([0.03, 1.61], device=dev)
    wiener = td.Wiener(dev, (w, h), overlap_factor=4, tile_size=32)
    bil = td.Bilateral(dev, (w, h), sigma_s=2.0, sigma_r=0.2)
    lum1, ab1 = wiener.process_log_luminance_lab(normalize_image(rgb, bounds), 0.075)
    lum2, ab2 = wiener.process_log_luminance_lab(rgb, 0.075, bounds=bounds)
    assert torch.equal(lum1, lum2) and torch.equal(ab1, ab2)
    x16 = rgb.half()
    out_a = bil.process_lab(*wiener.process_log_luminance_lab(normalize_image(x16, bounds), 0.075), 0.4, out_dtype=torch.float16)
    out_b = bil.process_lab(*wiener.process_log_luminance_lab(x16, 0.075, bounds=bounds), 0.4, out_dtype=torch.float16)
    rel = ((out_a.float() - out_b.float()).abs() / out_a.float().amax(-1, keepdim=True).clamp_min(0.05)).max().item()
    assert rel <= 1.5e-3, rel
    with pytest.raises(RuntimeError):
        wiener.process_log_luminance_lab(rgb, 0.075, bounds=torch.tensor([0.0, 1.0]))  # bounds on the host


def test_lab_chain_argument_checks(td, dev):
    wiener = td.Wiener(dev, (64, 64))
    bil = td.Bilateral(dev, (64, 64), sigma_s=2.0, sigma_r=0.2)
    x = torch.rand(64, 64, 3, device=dev)
    lum, ab = wiener.process_log_luminance_lab(x, 0.05)
    with pytest.raises(RuntimeError):
        bil.process_lab(lum.half(), ab, 0.4)
    with pytest.raises(RuntimeError):
        bil.process_lab(lum, ab[:, :, :1], 0.4)
    with pytest.raises(RuntimeError):
        wiener.process_log_luminance_lab(x[:32], 0.05)
    with pytest.raises(RuntimeError):
        wiener.process_log_luminance_lab(x, 0.05, chroma_out=torch.empty(64, 64, 2, device=dev, dtype=torch.float16))

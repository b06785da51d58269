"""CPU-only: host-side logic of the drop-in package (no kernel launches)."""

from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = np.load(Path(__file__).parent / 'golden' / 'reference_helpers.npz')


def test_public_surface_matches_reference_names(td):
    # reference torch_darktable/__init__.py:55-114
    expected = ['PPG', 'RCD', 'BayerPattern', 'Bilateral', 'Bilinear5x5', 'InputFormat', 'Jpeg', 'JpegException', 'Laplacian',
                'LaplacianParams', 'PackedFormat', 'PostProcess', 'Subsampling', 'TonemapParameters', 'Wiener', 'aces_tonemap',
                'apply_white_balance', 'bilinear5x5_demosaic', 'color_transform_3x3', 'compute_image_bounds', 'compute_image_metrics',
                'compute_log_luminance', 'compute_luminance', 'decode12', 'decode12_float', 'decode12_half', 'decode12_u16', 'encode',
                'encode12_float', 'encode12_u16', 'estimate_channel_noise', 'estimate_white_balance', 'lab_to_rgb', 'lab_to_xyz',
                'linear_tonemap', 'load_as_bayer', 'metrics_from_dict', 'metrics_to_dict', 'modify_hsl', 'modify_log_luminance',
                'modify_luminance', 'modify_vibrance', 'print_metrics', 'reinhard_tonemap', 'rgb_to_bayer', 'rgb_to_lab', 'rgb_to_xyz',
                'xyz_to_lab', 'xyz_to_rgb']
    for name in expected:
        assert hasattr(td, name), name
    ext = td.extension.extension
    for name in ('PPG', 'RCD', 'PostProcess', 'Laplacian', 'Bilateral', 'Wiener', 'TonemapParams', 'BayerPattern', 'adaptive_aces_tonemap',
                 'compute_image_bounds', 'apply_white_balance', 'estimate_white_balance', 'Jpeg', 'JpegInputFormat', 'JpegSubsampling'):
        assert hasattr(ext, name), name
    assert int(ext.BayerPattern.RGGB) == 0x94949494 and td.BayerPattern.GBRG.value == ext.BayerPattern.GBRG


@pytest.mark.parametrize('pat', ['RGGB', 'BGGR', 'GRBG', 'GBRG'])
def test_rgb_to_bayer_against_reference_fixture(td, pat):
    for tag in ('small', 'mid'):
        got = td.rgb_to_bayer(torch.from_numpy(GOLD[f'rgb_{tag}']), td.BayerPattern[pat]).numpy()
        assert np.array_equal(got, GOLD[f'bayer_{tag}_{pat}'])
    assert td.bayer.pixel_order(td.BayerPattern[pat]) == tuple(GOLD[f'pixel_order_{pat}'])
    assert td.bayer.channels(td.BayerPattern[pat]) == tuple(GOLD[f'channels_{pat}'])


def test_stack_expand_roundtrip(td):
    got = td.bayer.stack_bayer(td.rgb_to_bayer(torch.from_numpy(GOLD['rgb_small']))[:, :, 0]).numpy()
    assert np.array_equal(got, GOLD['stack_small'])
    x = torch.rand(6, 10)
    assert torch.equal(td.bayer.expand_bayer(td.bayer.stack_bayer(x))[:, :, 0], x)


@pytest.mark.parametrize('stride', [1, 8])
def test_estimate_channel_noise_against_reference_fixture(td, stride):
    got = td.estimate_channel_noise(torch.from_numpy(GOLD['noise_img']), stride).numpy()
    assert np.allclose(got, GOLD[f'noise_sigma_stride{stride}'], rtol=1e-6, atol=1e-8)


def test_no_cpu_fallback(td):
    """Like the reference, every kernel op refuses CPU tensors (no silent host path)."""
    with pytest.raises(RuntimeError):
        td.bilinear5x5_demosaic(torch.zeros(8, 8, 1), td.BayerPattern.RGGB)
    with pytest.raises(RuntimeError):
        td.rgb_to_lab(torch.zeros(8, 8, 3))
    with pytest.raises(RuntimeError):
        td.decode12_float(torch.zeros(6, dtype=torch.uint8))
    with pytest.raises(RuntimeError):
        td.compute_image_bounds([torch.zeros(8, 8, 3)], 8)
    with pytest.raises((RuntimeError, AssertionError)):
        td.reinhard_tonemap(torch.zeros(8, 8, 3), torch.zeros(5), td.TonemapParameters())
    with pytest.raises(ValueError):
        td.Wiener(torch.device('cpu'), (64, 64))
    with pytest.raises(RuntimeError):
        td.RCD(torch.device('cpu'), (64, 64), td.BayerPattern.RGGB)


def test_wrapper_validation(td):
    dev = torch.device('cuda', 0)  # constructing workspaces does not touch the GPU
    with pytest.raises(ValueError):
        td.Wiener(dev, (64, 64), overlap_factor=3)
    with pytest.raises(ValueError):
        td.Wiener(dev, (64, 64), tile_size=24)
    with pytest.raises(ValueError):
        td.Wiener(dev, (0, 64))
    w = td.Wiener(dev, (64, 48), overlap_factor=2, tile_size=16)
    assert w.overlap_factor == 2 and 'Wiener(64x48' in repr(w)
    with pytest.raises(RuntimeError):
        w.process(torch.zeros(48, 60, 1), 0.1)
    with pytest.raises(ValueError):
        w.process(torch.zeros(48, 64, 2), 0.1)
    ppg = td.PPG(dev, (64, 48), td.BayerPattern.GRBG, median_threshold=1.0)
    assert ppg.image_size == (64, 48) and ppg.median_threshold == 1.0
    with pytest.raises(RuntimeError):
        ppg.process(torch.zeros(48, 64))
    pp = td.PostProcess(dev, (64, 48), td.BayerPattern.RGGB, color_smoothing_passes=3, green_eq_threshold=0.1)
    assert pp.color_smoothing_passes == 3 and abs(pp.green_eq_threshold - 0.1) < 1e-9
    with pytest.raises(RuntimeError):
        td.Laplacian(dev, (64, 48), td.LaplacianParams(num_gamma=8))
    bil = td.Bilateral(dev, (64, 48), sigma_s=2.0, sigma_r=0.2)
    assert (bil.sigma_s, bil.sigma_r, bil.image_size) == (2.0, 0.2, (64, 48))
    with pytest.raises(ValueError):
        td.decode12(torch.zeros(3, dtype=torch.uint8), torch.int32)
    with pytest.raises(ValueError):
        td.encode(torch.zeros(2, dtype=torch.int32))


def test_tonemap_parameters_and_metrics_dict(td):
    p = td.TonemapParameters(0.75, 2.0, 1.0, 0.1)
    cpp = p.to_cpp()
    assert (cpp.gamma, cpp.intensity, cpp.light_adapt, cpp.vibrance) == (0.75, 2.0, 1.0, 0.1)
    assert td.TonemapParameters.from_cpp(cpp) == p
    assert td.TonemapParameters().light_adapt == 0.8 and td.extension.extension.TonemapParams().light_adapt == 0.8
    m = torch.tensor([-1.0, 0.4, 0.3, 0.5, 0.2])
    d = td.metrics_to_dict(m)
    assert d['log_mean'] == -1.0 and d['rgb_mean'] == pytest.approx((0.3, 0.5, 0.2))
    assert torch.allclose(td.metrics_from_dict(d, torch.device('cpu')), m)


def test_config1_pure_torch_bilinear_matches_oracle(oracle, scene):
    """BASELINE config 1: one 512 x 512 RGGB frame through a pure-torch CPU bilinear demosaic
    (test-side restatement of csrc/debayer/bilinear.cu: 13-tap diamond, replicate padding)
    against the C oracle.  Plumbing only -- the product has no CPU path."""
    bayer = oracle.mosaic(scene(512, 512, 1), oracle.RGGB)
    ref = oracle.bilinear5x5(bayer, oracle.RGGB)
    x = torch.from_numpy(bayer[:, :, 0])[None, None]
    xp = torch.nn.functional.pad(x, (2, 2, 2, 2), mode='replicate')
    k = {'ident': [[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 16, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]],
         'g_rb': [[0, 0, -2, 0, 0], [0, 0, 4, 0, 0], [-2, 4, 8, 4, -2], [0, 0, 4, 0, 0], [0, 0, -2, 0, 0]],
         'rb_br': [[0, 0, -3, 0, 0], [0, 4, 0, 4, 0], [-3, 0, 12, 0, -3], [0, 4, 0, 4, 0], [0, 0, -3, 0, 0]],
         'c_h': [[0, 0, 1, 0, 0], [0, -2, 0, -2, 0], [-2, 8, 10, 8, -2], [0, -2, 0, -2, 0], [0, 0, 1, 0, 0]],
         'c_v': [[0, 0, -2, 0, 0], [0, -2, 8, -2, 0], [1, 0, 10, 0, 1], [0, -2, 8, -2, 0], [0, 0, -2, 0, 0]]}
    resp = {n: torch.nn.functional.conv2d(xp, torch.tensor(v, dtype=torch.float32)[None, None])[0, 0] / 16 for n, v in k.items()}
    out = torch.empty(512, 512, 3)
    table = {(0, 0): ('ident', 'g_rb', 'rb_br'), (0, 1): ('c_h', 'ident', 'c_v'), (1, 0): ('c_v', 'ident', 'c_h'), (1, 1): ('rb_br', 'g_rb', 'ident')}
    for (r, c), names in table.items():
        for ch, n in enumerate(names):
            out[r::2, c::2, ch] = resp[n][r::2, c::2]
    assert np.abs(out.numpy() - ref).max() < 2e-6  # conv2d accumulates in a different order


def test_jpeg_surface_and_host_checks(td):
    """Jpeg keeps the reference API; the encoder is a device encoder, so a CPU tensor is rejected like the reference rejects it
    (jpeg_encoder.cu:134: "Input image should be on CUDA device").  The encoder itself: tests/test_gpu_jpeg.py."""
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(96), indexing='ij')
    img = torch.stack(((xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256), -1).to(torch.uint8)
    enc = td.Jpeg()
    with pytest.raises(RuntimeError, match='CUDA'):
        enc.encode(img, quality=95, input_format=td.InputFormat.RGBI, subsampling=td.Subsampling.CSS_444)
    with pytest.raises(RuntimeError):
        enc.encode(img, 95, 9, td.Subsampling.CSS_444)
    assert int(td.InputFormat.RGBI) == 3 and int(td.Subsampling.CSS_GRAY) == 2 and repr(td.extension.extension.Jpeg()) == 'Jpeg'
    assert issubclass(td.JpegException, Exception)
